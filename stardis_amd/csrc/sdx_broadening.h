// sdx_broadening.h — per-(line, depth) scalar formulas of the reference's broadening module, shared by the dense
// broadening kernels and by the pre-pass when it generates line parameters on the fly.
#pragma once
#include "sdx_math.h"

namespace sdx {

// broadening.py scalar formulas (:32-66, :114-137, :193-229, :281-343, :420-473)
__device__ __forceinline__ double n_effective(int ion, double e_ion, double e_lev)
{
    return mul_rn(sqrt(kRydEnergy / sub_rn(e_ion, e_lev)), (double)ion);
}
__device__ __forceinline__ double gamma_linear_stark(double nu_, double nl_, double ne)
{
    const double a1 = (sub_rn(nu_, nl_) < 1.5) ? 0.642 : 1.0;
    return mul_rn(mul_rn(mul_rn(0.60, a1), sub_rn(mul_rn(nu_, nu_), mul_rn(nl_, nl_))), pow(ne, 2.0 / 3.0));
}
__device__ __forceinline__ double gamma_quadratic_stark(int ion, double nu_, double nl_, double ne, double t)
{
    const double eps0 = 1.0 / (4.0 * kPi);
    const double zi = (double)ion;
    const double pre = mul_rn(mul_rn(mul_rn(mul_rn(kEesu, kEesu), kBohr), kBohr), kBohr) /
                       mul_rn(mul_rn(mul_rn(mul_rn(mul_rn(mul_rn(36.0, kH), eps0), zi), zi), zi), zi);
    const double t1 = mul_rn(nu_, add_rn(mul_rn(mul_rn(5.0, nu_), nu_), 1.0));
    const double t2 = mul_rn(nl_, add_rn(mul_rn(mul_rn(5.0, nl_), nl_), 1.0));
    const double c4 = mul_rn(pre, sub_rn(mul_rn(t1, t1), mul_rn(t2, t2)));
    return mul_rn(mul_rn(mul_rn(mul_rn(1e19, kKB), ne), pow(c4, 2.0 / 3.0)), pow(t, 1.0 / 6.0));
}
__device__ __forceinline__ double gamma_van_der_waals(int ion, double nu_, double nl_, double t, double nh)
{
    const double u2 = mul_rn(nu_, nu_), l2 = mul_rn(nl_, nl_);
    const double c6 = mul_rn(6.46e-34, sub_rn(add_rn(mul_rn(5.0, mul_rn(u2, u2)), u2), add_rn(mul_rn(5.0, mul_rn(l2, l2)), l2))) /
                      (double)(2 * ion * ion);
    return mul_rn(mul_rn(mul_rn(17.0, pow(mul_rn(mul_rn(8.0, kKB), t) / mul_rn(kPi, kMp), 0.3)), pow(c6, 0.4)), nh);
}
__device__ __forceinline__ double doppler_width(double nu, double t, double mass, double xi)
{
    return mul_rn(nu / kC, sqrt(add_rn(mul_rn(mul_rn(2.0, kKB), t) / mass, mul_rn(xi, xi))));
}

// (vbar / 1e6) ** (1 - alpha) of the ABO van der Waals width (broadening.py:939-948), vbar = sqrt((8 k T / pi) (1/mu)): the power as
// exp((1 - alpha) (0.5 (log(8 k T / pi) + log(1 / mu)) - log 1e6)) — the two logarithms depend on the depth alone and on the line
// alone, so an item costs one inline exponential where the reference's form costs a square root, a division and a pow.  Against
// that form: the logarithms are ~27 and ~55 in magnitude, their half-sum is within 1e-14 of the exact exponent, the result within
// ~1e-14 relative (tests: <= 1e-13 on gamma against the reference's arrays).  The dense kernel (vald_vdw) and the generating
// pre-pass (gen_gamma) share this function: they agree bit for bit.
constexpr double kLn1e6 = 0x1.ba18a998fffa0p+3;  // log(1e6), correctly rounded
__device__ __forceinline__ double exp_inline(double x);
__device__ __forceinline__ double abo_speed_power(double log_vb, double log_inv_mu, double one_minus_alpha)
{
    return exp_inline(mul_rn(one_minus_alpha, sub_rn(mul_rn(0.5, add_rn(log_vb, log_inv_mu)), kLn1e6)));
}

// broadening.py:880-1006
__device__ __forceinline__ double vald_stark(double ne, double stark, double t)
{
    const double g = mul_rn(mul_rn(ne, pow(10.0, stark)), pow(t / 1e4, 1.0 / 6));
    return (mul_rn(ne, stark) >= 0) ? 0.0 : g;
}
__device__ inline double vald_vdw(double vdw, double t, double mass, double e_up, double e_lo, double nh, int ion, double e_ion)
{
    double g = 0.0;
    if (vdw < 0) g = mul_rn(pow(10.0, vdw), pow(t / 1e4, 0.38));
    else if (vdw == 0.0) g = 0.0;
    else if (vdw < 20) {
        const double nu_ = n_effective(ion, e_ion, e_up);
        const double nl_ = n_effective(ion, e_ion, e_lo);
        g = mul_rn(gamma_van_der_waals(ion, nu_, nl_, t, 1.0), vdw);
    } else {
        const double vi = trunc(vdw);
        const double sigma = mul_rn(mul_rn(vi, kBohr), kBohr);
        const double alpha = sub_rn(vdw, vi);
        const double inv_mu = add_rn(1.0 / mul_rn(1.008, kAmu), 1.0 / mass);
        g = mul_rn(mul_rn(mul_rn(mul_rn(mul_rn(2.0, pow(4.0 / kPi, alpha / 2)), tgamma(sub_rn(4.0, alpha) / 2)), 1e6), sigma),
                   abo_speed_power(log(mul_rn(mul_rn(8.0, kKB), t) / kPi), log(inv_mu), sub_rn(1.0, alpha)));
    }
    return mul_rn(g, nh);
}


// ------------------------------------------------------------------------------------------------
// Line parameters generated from per-line scalars and per-depth state: what the reference tabulates
// as three dense (N_l, N_d) arrays before calc_alan_entries — alpha_line (plasma/base.py:200-321, :348-455;
// plasma/molecules.py:214-320, :345-440), gammas and doppler_widths (broadening.py:659-732, :735-821, :1009-1085) —
// evaluated for one (line, depth) item.  ~80 B per line cross HBM instead of 24 B x N_d.
struct LineParams {
    // alpha_line
    const double* e_low_ev;   // [N_l] lower level energy, eV
    const double* g_lo;       // [N_l] 2 j_lo + 1, or null for short lists
    const double* strength;   // [N_l] f_lu = 10**log_gf / g_lo, or 10**log_gf for short lists; null = dense inputs
    const int* pop_row;       // [N_l] row of `pop` for the line's ion / molecule
    const double* pop;        // [rows][N_d] number density / partition function
    double alpha_coefficient;
    // doppler width
    const double* mass;       // [N_l]
    double xi;                // microturbulence, cm/s
    // gamma
    int gamma_mode;           // 0 calc_gamma, 1 calc_vald_gamma, 2 A_ul only (molecules, one column), 3 zero
    int flags;                // 1 linear Stark, 2 quadratic Stark, 4 van der Waals, 8 radiation
    const int* z;
    const int* ion;           // charge seen by the outer electron (ion_number + 1)
    const double* e_ion;
    const double* e_up;
    const double* e_lo;
    const double* a_ul;
    const double* stark;
    const double* waals;
    // per depth
    const double* temps;
    const double* n_e;
    const double* n_h;
};

constexpr double kKBsi = 1.380649e-23;
constexpr double kHsi = 6.62607015e-34;
constexpr double kEvJ = 1.602176634e-19;

// The three values of one (line, depth) item are products of a factor that depends on the line only, a factor that
// depends on the depth only, and a few operations that need both.  Every transcendental that can be (pow, tgamma, the
// n_eff square roots) sits in the per-line or the per-depth part, which the pre-pass evaluates once per block row /
// column instead of once per item; the combination keeps the operation order of the scalar formulas above, so the
// result has the bits of gamma_linear_stark(), vald_vdw() & co. evaluated directly.
struct GenDepth {
    double t, ne, nh;
    double inv_kt;   // 1 / (T k_B[SI])                        alpha: Boltzmann factor
    double inv_t;    // 1 / T                                  alpha: stimulated-emission correction
    double two_kt;   // 2 k_B T                                Doppler width
    double ne23;     // n_e ** (2/3)                           linear Stark
    double t16;      // (T / 1e4) ** (1/6) [VALD] or T ** (1/6) * ... see gen_item
    double t38;      // (T / 1e4) ** 0.38                      VALD van der Waals, log form
    double vw;       // 17 (8 k T / (pi m_p)) ** 0.3           Unsoeld van der Waals
    double lvb;      // log(8 k T / pi)                        ABO van der Waals (abo_speed_power)
    double qd;       // 1e19 k_B n_e                           quadratic Stark
};
struct GenLine {
    double nu_over_c;  // Doppler width
    double lin;        // 0.6 a1 (n_u^2 - n_l^2), or 0 when the line takes no linear Stark term
    double c4p;        // c4 ** (2/3)                          quadratic Stark
    double c6p;        // c6 ** 0.4                            Unsoeld van der Waals
    double p10s;       // 10 ** stark
    double p10w;       // 10 ** waals (log form)
    double ab;         // 2 (4/pi)**(alpha/2) Gamma((4-alpha)/2) 1e6 sigma   (ABO)
    double oma;        // 1 - alpha                                           (ABO)
    double lmu;        // log(1/(1.008 amu) + 1/m)                            (ABO)
    // the line's own scalars, read once per line instead of once per item (a pre-pass block keeps its lines' GenLine in LDS)
    double nu, e_low_ev, g_lo, strength, mass, inv_mass, a_ul, stark, waals;
    int pop_row, is_h;  // is_h: atomic number 1 (the line takes the linear Stark term)
};

__device__ inline GenDepth gen_depth(const LineParams& p, int d)
{
    GenDepth D{};
    const double t = p.temps[d];
    D.t = t;
    D.inv_kt = 1.0 / mul_rn(t, kKBsi);
    D.inv_t = 1.0 / t;
    D.two_kt = mul_rn(mul_rn(2.0, kKB), t);
    if (p.gamma_mode > 1) return D;
    D.ne = p.n_e[d];
    D.nh = p.n_h[d];
    if (p.flags & 1) D.ne23 = pow(D.ne, 2.0 / 3.0);
    if (p.flags & 4) {
        D.vw = mul_rn(17.0, pow(mul_rn(mul_rn(8.0, kKB), t) / mul_rn(kPi, kMp), 0.3));
        if (p.gamma_mode == 1) D.lvb = log(mul_rn(mul_rn(8.0, kKB), t) / kPi);
    }
    if (p.gamma_mode == 0) {
        if (p.flags & 2) {
            D.t16 = pow(t, 1.0 / 6.0);
            D.qd = mul_rn(mul_rn(1e19, kKB), D.ne);
        }
    } else {
        if (p.flags & 2) D.t16 = pow(t / 1e4, 1.0 / 6);
        if (p.flags & 4) D.t38 = pow(t / 1e4, 0.38);
    }
    return D;
}

__device__ inline GenLine gen_line(const LineParams& p, double line_nu, int64_t l)
{
    GenLine L{};
    L.nu_over_c = line_nu / kC;
    L.nu = line_nu;
    if (p.strength) {  // (alpha from per-line scalars: gen_alpha)
        L.e_low_ev = p.e_low_ev[l];
        L.g_lo = p.g_lo ? p.g_lo[l] : 1.0;
        L.strength = p.strength[l];
        L.pop_row = p.pop_row[l];
    }
    L.mass = p.mass[l];
    L.inv_mass = 1.0 / L.mass;
    if (p.gamma_mode <= 2 && p.a_ul) L.a_ul = p.a_ul[l];
    if (p.gamma_mode > 1) return L;
    L.is_h = (p.flags & 1) && p.z[l] == 1;
    if (p.gamma_mode == 1) {
        if (p.flags & 2) L.stark = p.stark[l];
        if (p.flags & 4) L.waals = p.waals[l];
    }
    const int ion = p.ion[l];
    const double e_ion = p.e_ion[l], e_up = p.e_up[l], e_lo = p.e_lo[l];
    const bool vald = p.gamma_mode == 1;
    const double vdw = vald ? p.waals[l] : 0.0;
    const bool lin = (p.flags & 1) && p.z[l] == 1;
    const bool unsoeld = (p.flags & 4) && (!vald || (vdw > 0.0 && vdw < 20));
    const bool quad0 = !vald && (p.flags & 2);
    if (lin || unsoeld || quad0) {
        const double nu_ = n_effective(ion, e_ion, e_up);
        const double nl_ = n_effective(ion, e_ion, e_lo);
        if (lin) {
            const double a1 = (sub_rn(nu_, nl_) < 1.5) ? 0.642 : 1.0;
            L.lin = mul_rn(mul_rn(0.60, a1), sub_rn(mul_rn(nu_, nu_), mul_rn(nl_, nl_)));
        }
        if (quad0) {
            const double eps0 = 1.0 / (4.0 * kPi);
            const double zi = (double)ion;
            const double pre = mul_rn(mul_rn(mul_rn(mul_rn(kEesu, kEesu), kBohr), kBohr), kBohr) /
                               mul_rn(mul_rn(mul_rn(mul_rn(mul_rn(mul_rn(36.0, kH), eps0), zi), zi), zi), zi);
            const double t1 = mul_rn(nu_, add_rn(mul_rn(mul_rn(5.0, nu_), nu_), 1.0));
            const double t2 = mul_rn(nl_, add_rn(mul_rn(mul_rn(5.0, nl_), nl_), 1.0));
            L.c4p = pow(mul_rn(pre, sub_rn(mul_rn(t1, t1), mul_rn(t2, t2))), 2.0 / 3.0);
        }
        if (unsoeld) {
            const double u2 = mul_rn(nu_, nu_), l2 = mul_rn(nl_, nl_);
            const double c6 = mul_rn(6.46e-34, sub_rn(add_rn(mul_rn(5.0, mul_rn(u2, u2)), u2), add_rn(mul_rn(5.0, mul_rn(l2, l2)), l2))) /
                              (double)(2 * ion * ion);
            L.c6p = pow(c6, 0.4);
        }
    }
    if (vald) {
        if (p.flags & 2) L.p10s = pow(10.0, p.stark[l]);
        if (p.flags & 4) {
            if (vdw < 0) L.p10w = pow(10.0, vdw);
            else if (vdw >= 20) {
                const double vi = trunc(vdw);
                const double sigma = mul_rn(mul_rn(vi, kBohr), kBohr);
                const double alpha = sub_rn(vdw, vi);
                L.lmu = log(add_rn(1.0 / mul_rn(1.008, kAmu), 1.0 / p.mass[l]));
                L.ab = mul_rn(mul_rn(mul_rn(mul_rn(2.0, pow(4.0 / kPi, alpha / 2)), tgamma(sub_rn(4.0, alpha) / 2)), 1e6), sigma);
                L.oma = sub_rn(1.0, alpha);
            }
        }
    }
    return L;
}

// exp(x) for the arguments an item meets (-E / kT, -h nu / kT, the ABO exponent: |x| < 700): the device library's sequence written out
// (exp_neg: the same constants and FMAs, hence the same bits) with its coefficients in scalar registers — ~25 instructions and no call
// (rounds 4 - 5 called the library's exp out of line: inlined, it pushed the pre-pass past its 64 VGPRs).  Round 6: the two calls per item were a fifth of the
// generating pre-pass's 1020 instructions per (line, depth).
__device__ __forceinline__ double exp_inline(double x)
{
    // (the library's own range handling around the same core: 0 below the smallest subnormal, inf above the largest double; NaN passes)
    const double r = exp_neg(-x);
    return x < -745.2 ? 0.0 : (x > 709.8 ? INFINITY : r);
}
// a / b, correctly rounded, from the correctly rounded reciprocal rb = 1 / b (Markstein: q = a rb is within an ulp, the residual
// a - q b is exact in one FMA, one correction step rounds correctly): three instructions where the division sequence takes ~25.
// Non-finite or overflowing quotients take the division itself.
__device__ __forceinline__ double div_by(double a, double b, double rb)
{
    const double q = mul_rn(a, rb);
    const double e = fma(-q, b, a);
    const double r = fma(e, rb, q);
    return (fabs(q) < 1e300 && fabs(q) > 1e-290) ? r : a / b;
}

__device__ inline double gen_alpha(const LineParams& p, const GenLine& L, const GenDepth& D, int d, int n_depth)
{
    const double expo = exp_inline(mul_rn(mul_rn(-L.e_low_ev, D.inv_kt), kEvJ));               // base.py:247-251
    double n_lower = mul_rn(expo, p.pop[(size_t)L.pop_row * n_depth + d]);                     // :254-266
    if (p.g_lo) n_lower = mul_rn(n_lower, L.g_lo);
    const double corr = sub_rn(1.0, exp_inline(mul_rn((-kHsi) / kKBsi, mul_rn(L.nu, D.inv_t))));  // :276-286
    return mul_rn(mul_rn(mul_rn(p.alpha_coefficient, n_lower), L.strength), corr);             // :288-296
}

__device__ inline double gen_doppler(const LineParams& p, const GenLine& L, const GenDepth& D)
{
    // (2 k T / m by div_by's three instructions without its fall-back: T and m are ordinary positive magnitudes — a zero mass or
    // temperature never reaches the generating pre-pass, the host refuses it)
    const double q = mul_rn(D.two_kt, L.inv_mass);
    const double two_kt_over_m = fma(fma(-q, L.mass, D.two_kt), L.inv_mass, q);
    return mul_rn(L.nu_over_c, sqrt(add_rn(two_kt_over_m, mul_rn(p.xi, p.xi))));  // broadening.py:32-66
}

__device__ inline double gen_gamma(const LineParams& p, const GenLine& L, const GenDepth& D)
{
    if (p.gamma_mode == 2) return L.a_ul;  // broadening.py:799-801
    if (p.gamma_mode == 3) return 0.0;
    const bool lin = (p.flags & 1) && L.is_h;
    const double g_lin = lin ? mul_rn(L.lin, D.ne23) : 0.0;  // :193-229
    if (p.gamma_mode == 0) {                                   // :550-656
        const double g_q = (p.flags & 2) ? mul_rn(mul_rn(D.qd, L.c4p), D.t16) : 0.0;   // :281-343
        const double g_w = (p.flags & 4) ? mul_rn(mul_rn(D.vw, L.c6p), D.nh) : 0.0;    // :420-473
        const double g_r = (p.flags & 8) ? L.a_ul : 0.0;
        return add_rn(add_rn(add_rn(g_lin, g_q), g_w), g_r);
    }
    double g = 0.0;  // :1009-1085
    if (p.flags & 8) g = add_rn(g, L.a_ul);
    if (lin) g = add_rn(g, g_lin);
    if (p.flags & 2) {  // :880-910
        const double gs = mul_rn(mul_rn(D.ne, L.p10s), D.t16);
        g = add_rn(g, (mul_rn(D.ne, L.stark) >= 0) ? 0.0 : gs);
    }
    if (p.flags & 4) {  // :913-1006
        const double vdw = L.waals;
        double gw = 0.0;
        if (vdw < 0) gw = mul_rn(L.p10w, D.t38);
        else if (vdw == 0.0) gw = 0.0;
        else if (vdw < 20) gw = mul_rn(mul_rn(mul_rn(D.vw, L.c6p), 1.0), vdw);
        else gw = mul_rn(L.ab, abo_speed_power(D.lvb, L.lmu, L.oma));
        g = add_rn(g, mul_rn(gw, D.nh));
    }
    return g / 2;
}

}  // namespace sdx
