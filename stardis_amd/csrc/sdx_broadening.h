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
        const double vbar = sqrt(mul_rn(mul_rn(mul_rn(8.0, kKB), t) / kPi, inv_mu));
        g = mul_rn(mul_rn(mul_rn(mul_rn(mul_rn(2.0, pow(4.0 / kPi, alpha / 2)), tgamma(sub_rn(4.0, alpha) / 2)), 1e6), sigma),
                   pow(vbar / 1e6, sub_rn(1.0, alpha)));
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
    double vb;       // 8 k T / pi                             ABO van der Waals (mean speed)
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
    double inv_mu;     // 1/(1.008 amu) + 1/m                                 (ABO)
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
        D.vb = mul_rn(mul_rn(8.0, kKB), t) / kPi;
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
    if (p.gamma_mode > 1) return L;
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
                L.inv_mu = add_rn(1.0 / mul_rn(1.008, kAmu), 1.0 / p.mass[l]);
                L.ab = mul_rn(mul_rn(mul_rn(mul_rn(2.0, pow(4.0 / kPi, alpha / 2)), tgamma(sub_rn(4.0, alpha) / 2)), 1e6), sigma);
                L.oma = sub_rn(1.0, alpha);
            }
        }
    }
    return L;
}

// out-of-line library calls for the pre-pass: inlined, pow / exp push the generating pre-pass past 64 VGPRs, and beyond
// that only ONE 1024-thread block fits a CU
__device__ __attribute__((noinline)) double pow_call(double a, double b) { return pow(a, b); }
__device__ __attribute__((noinline)) double exp_call(double a) { return exp(a); }

__device__ inline double gen_alpha(const LineParams& p, const GenDepth& D, double line_nu, int64_t l, int d, int n_depth)
{
    const double expo = exp_call(mul_rn(mul_rn(-p.e_low_ev[l], D.inv_kt), kEvJ));             // base.py:247-251
    double n_lower = mul_rn(expo, p.pop[(size_t)p.pop_row[l] * n_depth + d]);                  // :254-266
    if (p.g_lo) n_lower = mul_rn(n_lower, p.g_lo[l]);
    const double corr = sub_rn(1.0, exp_call(mul_rn((-kHsi) / kKBsi, mul_rn(line_nu, D.inv_t))));  // :276-286
    return mul_rn(mul_rn(mul_rn(p.alpha_coefficient, n_lower), p.strength[l]), corr);          // :288-296
}

__device__ inline double gen_doppler(const LineParams& p, const GenLine& L, const GenDepth& D, int64_t l)
{
    return mul_rn(L.nu_over_c, sqrt(add_rn(D.two_kt / p.mass[l], mul_rn(p.xi, p.xi))));  // broadening.py:32-66
}

__device__ inline double gen_gamma(const LineParams& p, const GenLine& L, const GenDepth& D, int64_t l)
{
    if (p.gamma_mode == 2) return p.a_ul[l];  // broadening.py:799-801
    if (p.gamma_mode == 3) return 0.0;
    const bool lin = (p.flags & 1) && p.z[l] == 1;
    const double g_lin = lin ? mul_rn(L.lin, D.ne23) : 0.0;  // :193-229
    if (p.gamma_mode == 0) {                                   // :550-656
        const double g_q = (p.flags & 2) ? mul_rn(mul_rn(D.qd, L.c4p), D.t16) : 0.0;   // :281-343
        const double g_w = (p.flags & 4) ? mul_rn(mul_rn(D.vw, L.c6p), D.nh) : 0.0;    // :420-473
        const double g_r = (p.flags & 8) ? p.a_ul[l] : 0.0;
        return add_rn(add_rn(add_rn(g_lin, g_q), g_w), g_r);
    }
    double g = 0.0;  // :1009-1085
    if (p.flags & 8) g = add_rn(g, p.a_ul[l]);
    if (lin) g = add_rn(g, g_lin);
    if (p.flags & 2) {  // :880-910
        const double gs = mul_rn(mul_rn(D.ne, L.p10s), D.t16);
        g = add_rn(g, (mul_rn(D.ne, p.stark[l]) >= 0) ? 0.0 : gs);
    }
    if (p.flags & 4) {  // :913-1006
        const double vdw = p.waals[l];
        double gw = 0.0;
        if (vdw < 0) gw = mul_rn(L.p10w, D.t38);
        else if (vdw == 0.0) gw = 0.0;
        else if (vdw < 20) gw = mul_rn(mul_rn(mul_rn(D.vw, L.c6p), 1.0), vdw);
        else gw = mul_rn(L.ab, pow_call(sqrt(mul_rn(D.vb, L.inv_mu)) / 1e6, L.oma));
        g = add_rn(g, mul_rn(gw, D.nh));
    }
    return g / 2;
}

}  // namespace sdx
