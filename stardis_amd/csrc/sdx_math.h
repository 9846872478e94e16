// sdx_math.h — fp64 device math for the STARDIS hot path on gfx950 (CDNA4).
//
// Two flavours of the Humlicek-W4 Faddeeva approximation the reference uses
// (stardis/radiation_field/opacities/opacities_solvers/voigt.py:17-86):
//
//   faddeeva_full()   complex in / complex out, the same operation order as the reference
//                     (Smith complex division, no FMA) — backs the element-wise
//                     faddeeva / voigt_profile entry points.
//   voigt_term()      what the line-opacity kernel evaluates per (line, depth, nu):
//                     amp * Re w(z), real part only, FMA, one reciprocal per point.
//                     Region tests are the reference's predicates (voigt.py:31-44) on its y and on
//                     x = delta_nu * (1 / doppler), which equals the reference's delta_nu / doppler
//                     to 1 ulp: a point within an ulp of a region boundary (s = 15, s = 5.5,
//                     y = 0.195 |x| - 0.176) can be evaluated with the neighbouring region's rational,
//                     a ~1e-5 step of the Humlicek approximation itself.  Inside a region the
//                     arithmetic is re-associated (few-ulp level; tests/test_gpu_hot_faddeeva.py pins
//                     the routine point by point against the reference's vectors: <= 2e-13).
//
// Everything is IEEE double; nothing here uses fast-math.
#pragma once
#include <hip/hip_runtime.h>

namespace sdx {

constexpr double kPi = 3.141592653589793;
constexpr double kSqrtPi = 1.7724538509055159;      // np.sqrt(np.pi), voigt.py:12
constexpr double kInvSqrtPi = 0.5641895835477563;   // 1.0 / kSqrtPi, correctly rounded
constexpr double kH = 6.62607015e-27;
constexpr double kC = 29979245800.0;
constexpr double kKB = 1.380649e-16;
constexpr double kMp = 1.67262192369e-24;
constexpr double kAmu = 1.6605390666e-24;
constexpr double kEesu = 4.803204712570263e-10;
constexpr double kBohr = 5.2917721090299995e-09;
constexpr double kSigmaT = 6.6524587321000005e-25;
constexpr double kRydFreq = 3289841960250881.0;
constexpr double kRydEnergy = 2.1798723611035848e-11;
constexpr double kBfConst = 2.815403624709817e+29;
constexpr double kFfConst = 369234910.67735106;
constexpr double kRydCm = 109737.31568160003;

struct c64 {
    double re, im;
};

// ---- exact-rounding helpers: the compiler may not contract these into FMAs -------------
__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ double sub_rn(double a, double b) { return __dsub_rn(a, b); }

// ---- reference-order complex arithmetic (CPython _Py_c_prod / _Py_c_quot) ----------------
__device__ __forceinline__ c64 cmul_ref(c64 a, c64 b)
{
    return {sub_rn(mul_rn(a.re, b.re), mul_rn(a.im, b.im)), add_rn(mul_rn(a.re, b.im), mul_rn(a.im, b.re))};
}
__device__ __forceinline__ c64 cdiv_ref(c64 a, c64 b)
{
    const double abr = fabs(b.re), abi = fabs(b.im);
    c64 r;
    if (abr >= abi) {
        const double ratio = b.im / b.re;
        const double denom = add_rn(b.re, mul_rn(b.im, ratio));
        r.re = add_rn(a.re, mul_rn(a.im, ratio)) / denom;
        r.im = sub_rn(a.im, mul_rn(a.re, ratio)) / denom;
    } else {
        const double ratio = b.re / b.im;
        const double denom = add_rn(mul_rn(b.re, ratio), b.im);
        r.re = add_rn(mul_rn(a.re, ratio), a.im) / denom;
        r.im = sub_rn(mul_rn(a.im, ratio), a.re) / denom;
    }
    return r;
}
__device__ __forceinline__ c64 fadd_c(double f, c64 z) { return {add_rn(f, z.re), z.im}; }
__device__ __forceinline__ c64 fsub_c(double f, c64 z) { return {sub_rn(f, z.re), -z.im}; }

// voigt.py:17-86, complex result.
__device__ inline c64 faddeeva_full(c64 z)
{
    const double x = z.re, y = z.im;
    const c64 t = {y, -x};
    const double s = add_rn(fabs(x), y);
    if (s > 15.0) {  // region I  (:47)
        const c64 a = {-mul_rn(kInvSqrtPi, y), mul_rn(kInvSqrtPi, x)};
        c64 z2 = cmul_ref(z, z);
        z2.re = sub_rn(z2.re, 0.5);
        return cdiv_ref(a, z2);
    }
    if (s > 5.5) {  // region II (:50-56)
        const c64 z2 = cmul_ref(z, z);
        const c64 inner = {sub_rn(z2.re / kSqrtPi, 1.4104739589), z2.im / kSqrtPi};
        const c64 zi = cmul_ref(z, inner);
        const c64 num = {-zi.im, zi.re};  // 1j * (...)
        c64 q = {sub_rn(z2.re, 3.0), z2.im};
        q = cmul_ref(z2, q);
        q.re = add_rn(0.75, q.re);
        return cdiv_ref(num, q);
    }
    if (y >= sub_rn(mul_rn(0.195, fabs(x)), 0.176)) {  // region III (:59-67)
        c64 p = fadd_c(3.778987, c64{mul_rn(0.5642236, t.re), mul_rn(0.5642236, t.im)});
        p = fadd_c(11.96482, cmul_ref(t, p));
        p = fadd_c(20.20933, cmul_ref(t, p));
        p = fadd_c(16.4955, cmul_ref(t, p));
        c64 q = fadd_c(6.699398, t);
        q = fadd_c(21.69274, cmul_ref(t, q));
        q = fadd_c(39.27121, cmul_ref(t, q));
        q = fadd_c(38.82363, cmul_ref(t, q));
        q = fadd_c(16.4955, cmul_ref(t, q));
        return cdiv_ref(p, q);
    }
    // region IV (:70-84)
    const c64 u = cmul_ref(t, t);
    c64 p = fsub_c(1.320522, c64{mul_rn(u.re, 0.56419), mul_rn(u.im, 0.56419)});
    p = fsub_c(35.7668, cmul_ref(u, p));
    p = fsub_c(219.031, cmul_ref(u, p));
    p = fsub_c(1540.787, cmul_ref(u, p));
    p = fsub_c(3321.99, cmul_ref(u, p));
    p = fsub_c(36183.31, cmul_ref(u, p));
    const c64 num = cmul_ref(t, p);
    c64 q = fsub_c(1.84144, u);
    q = fsub_c(61.5704, cmul_ref(u, q));
    q = fsub_c(364.219, cmul_ref(u, q));
    q = fsub_c(2186.18, cmul_ref(u, q));
    q = fsub_c(9022.23, cmul_ref(u, q));
    q = fsub_c(24322.8, cmul_ref(u, q));
    q = fsub_c(32066.6, cmul_ref(u, q));
    const c64 frac = cdiv_ref(num, q);
    const double l = exp(u.re);
    double sn, cs;
    sincos(u.im, &sn, &cs);
    return {sub_rn(mul_rn(l, cs), frac.re), sub_rn(mul_rn(l, sn), frac.im)};
}

// voigt.py:113-150
__device__ inline double voigt_profile_full(double delta_nu, double doppler_width, double gamma)
{
    const c64 z = {delta_nu / doppler_width, (gamma / mul_rn(kSqrtPi, kPi)) / doppler_width};
    return faddeeva_full(z).re / mul_rn(kSqrtPi, doppler_width);
}

// ---- fast path ---------------------------------------------------------------------------
__device__ __forceinline__ c64 cmul(c64 a, c64 b)
{
    return {fma(a.re, b.re, -(a.im * b.im)), fma(a.re, b.im, a.im * b.re)};
}
// f + t*p and f - u*p, Horner steps with a real constant
__device__ __forceinline__ c64 horner_add(double f, c64 t, c64 p)
{
    return {fma(t.re, p.re, fma(-t.im, p.im, f)), fma(t.re, p.im, t.im * p.re)};
}
__device__ __forceinline__ c64 horner_sub(double f, c64 u, c64 p)
{
    return {fma(-u.re, p.re, fma(u.im, p.im, f)), -fma(u.re, p.im, u.im * p.re)};
}
// 1/d from the hardware seed (good to 4.6e-8) and ONE cubically convergent step, r (1 + e + e^2) with e = 1 - d r:
// the truncation error is e^3 ~ 1e-22, so the result is the correctly rounded quotient except in ~1 of 1e8 cases where it
// is the neighbouring double (scripts/recip_check.hip) — one instruction fewer than two Newton steps.  No denormal /
// range fix-up: callers pass |z|^2-like magnitudes far from the exponent limits.
__device__ __forceinline__ double recip(double d)
{
    const double r = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r, 1.0);
    return fma(r, fma(e, e, e), r);
}

// acc + amp * Re w(x + i y) for region I:  Re[ (i/sqrt(pi)) z / (z^2 - 1/2) ]  (voigt.py:47)
//   = (y/sqrt(pi)) (q + y^2 + 1/2) / ((q - y^2 - 1/2)^2 + 4 q y^2),  q = x^2.
// With v = q + y^2 - 1/2 the denominator is v^2 + 2 y^2 and the numerator's bracket is v + 1: no cancellation anywhere
// (v >= 100 in region I), and the point costs TEN instructions — v = fma(x, x, cv), den = fma(v, v, cd),
// num = fma(yk, v, yk), the reciprocal (4) and fma(num, 1/den, acc); the difference nu - nu_l and the scaling by 1/dnu_D
// are the other two.  The sum is part of the routine: every caller adds a region-I term with this one FMA, so a point
// gets the same bits whichever path (test-free tile, edge block, core block, narrow role) evaluates it.
struct RegionI {
    double yk;   // amp * y / sqrt(pi): the line's amplitude is folded into the numerator
    double cv;   // y^2 - 1/2
    double cd;   // 2 y^2
};
__device__ __forceinline__ RegionI region1_setup(double y, double amp)
{
    const double y2 = y * y;
    return {amp * (y * kInvSqrtPi), y2 - 0.5, y2 + y2};
}
__device__ __forceinline__ double region1_add(double acc, double x, const RegionI& k)
{
    const double v = fma(x, x, k.cv);
    const double den = fma(v, v, k.cd);
    const double num = fma(k.yk, v, k.yk);
    const double r = recip(den);
    // acc <- fma(num, r, acc) IN PLACE: left to the compiler, the loop-carried sums of the line kernel are formed in fresh
    // registers and copied back (five v_mov_b64 per walked line next to forty useful instructions)
    asm("v_fma_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(num), "v"(r));
    return acc;
}
// R region-I terms of ONE line at the R grid points of a lane (a test-free tile of the wide role) with ONE refined reciprocal
// between them: r = 1 / (d0 d1 d2 d3), then 1/d0 = (r d2 d3) d1 ... — three products, the reciprocal (a quarter-rate
// instruction and its three refinement FMAs) and six more products where four separate reciprocals cost sixteen instructions,
// four of them quarter-rate.  x comes from the lane's frequency OFFSETS within the tile, x = fma(dnu, inv, c0) with
// c0 = (nu_base - nu_l) * inv formed once per (line, tile): one instruction per point instead of two.  The denominators are
// >= 1e4 (|x| + y > 15) and <= ~1e24 (a window spanning the whole optical grid): their product stays far inside the exponent
// range.  Each factor is within 1 ulp, the terms within ~3 ulp of region1_add's; which points share a reciprocal is fixed by
// the grid (global tiles), so the bits do not depend on sharding.
template <int R>
__device__ __forceinline__ void region1_add_shared(double (&acc)[R], const double (&dnu)[R], double inv, double c0, const RegionI& k)
{
    static_assert(R == 4 || R == 8, "four points share a reciprocal");
#pragma unroll
    for (int q = 0; q < R; q += 4) {
        double den[4], num[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double x = fma(dnu[q + j], inv, c0);
            const double v = fma(x, x, k.cv);
            den[j] = fma(v, v, k.cd);
            num[j] = fma(k.yk, v, k.yk);
        }
        const double p01 = den[0] * den[1], p23 = den[2] * den[3];
        const double r = recip(p01 * p23);
        const double r01 = r * p23, r23 = r * p01;  // 1 / (d0 d1), 1 / (d2 d3)
        const double i0 = r01 * den[1], i1 = r01 * den[0], i2 = r23 * den[3], i3 = r23 * den[2];
        asm("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[q + 0]) : "v"(num[0]), "v"(i0));
        asm("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[q + 1]) : "v"(num[1]), "v"(i1));
        asm("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[q + 2]) : "v"(num[2]), "v"(i2));
        asm("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[q + 3]) : "v"(num[3]), "v"(i3));
    }
}
// the same where only some lanes take the term: the others add num * 0 (num is finite whatever x is; the reciprocal may
// not be — v can vanish next to a line centre, where callers pass take = false)
__device__ __forceinline__ double region1_add_if(double acc, double x, const RegionI& k, bool take)
{
    const double v = fma(x, x, k.cv);
    const double den = fma(v, v, k.cd);
    const double num = fma(k.yk, v, k.yk);
    const double r = take ? recip(den) : 0.0;
    asm("v_fma_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(num), "v"(r));
    return acc;
}

// exp(-tau) for 0 <= tau < 700: the ROCm device library's double-precision exp, operation for operation (same
// constants, same FMA sequence, hence the same bits as exp(-tau)), written out so that every Horner step is ONE
// three-address v_fma_f64.  The compiler lowers the library's chain to two-address v_fmac_f64 plus a v_mov_b64 of the
// coefficient per step — nine extra issue slots in the innermost loop of the formal solution.
__device__ __forceinline__ double fma3(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ double exp_neg(double tau)
{
    const double n = rint(mul_rn(tau, -0x1.71547652b82fep+0));       // -log2(e)
    double r = fma(n, -0x1.62e42fefa39efp-1, -tau);                    // -ln2 (high part)
    r = fma(-0x1.abc9e3b39803fp-56, n, r);                             // -ln2 (low part)
    // the nine three-address Horner steps as ONE asm block: after every separate asm statement the compiler inserts a
    // defensive s_nop, which would hand back the issue slots the three-address form saves.  The coefficients sit in SGPR
    // pairs (one scalar operand per VALU instruction is allowed): as VGPR operands they would pin 20 vector registers
    // across the caller's whole loop; only the leading coefficient, the second constant of the first step, is a VGPR.
    double p;
    asm("v_fma_f64 %0, %2, %1, %3\n\t"
        "v_fma_f64 %0, %1, %0, %4\n\t"
        "v_fma_f64 %0, %1, %0, %5\n\t"
        "v_fma_f64 %0, %1, %0, %6\n\t"
        "v_fma_f64 %0, %1, %0, %7\n\t"
        "v_fma_f64 %0, %1, %0, %8\n\t"
        "v_fma_f64 %0, %1, %0, %9\n\t"
        "v_fma_f64 %0, %1, %0, %10\n\t"
        "v_fma_f64 %0, %1, %0, %11"
        : "=&v"(p)
        : "v"(r), "v"(0x1.ade156a5dcb37p-26), "s"(0x1.28af3fca7ab0cp-22), "s"(0x1.71dee623fde64p-19), "s"(0x1.a01997c89e6b0p-16),
          "s"(0x1.a01a014761f6ep-13), "s"(0x1.6c16c1852b7b0p-10), "s"(0x1.1111111122322p-7), "s"(0x1.55555555502a1p-5),
          "s"(0x1.5555555555511p-3), "s"(0x1.000000000000bp-1));
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    return ldexp(p, (int)n);
}

// cos(a) for |a| < ~1e3 (region IV of the Faddeeva function needs |a| = 2 |x| y < 10): two-term Cody-Waite reduction to
// |r| <= pi/4 and the fdlibm minimax polynomials for sin and cos on that interval (both evaluated: the quadrant differs
// from lane to lane), < 1 ulp.  The device library's cos carries an argument-reduction path for huge arguments and costs
// about three times as many instructions.
__device__ __forceinline__ double cos_small(double a)
{
    const double n = rint(a * 0x1.45f306dc9c883p-1);              // 2 / pi
#ifndef SDX_NO_COS_SHORT
    // |a| < pi/4 in every lane of the wave (the usual case: a = 2 x y with a small damping parameter y): no reduction, quadrant
    // 0, only the cosine polynomial — the same operations on the same operands as the general path takes for n = 0 (its two
    // reduction FMAs return a unchanged), hence the same bits, 14 instructions instead of 36
    if (__builtin_amdgcn_ballot_w64(n != 0.0) == 0) {
        const double z = a * a;
        double pc = fma(z, -0x1.8fae9be8838d4p-37, 0x1.1ee9ebdb4b1c4p-29);
        pc = fma(z, pc, -0x1.27e4f809c52adp-22);
        pc = fma(z, pc, 0x1.a01a019cb1590p-16);
        pc = fma(z, pc, -0x1.6c16c16c15177p-10);
        pc = fma(z, pc, 0x1.555555555554cp-5);
        const double hz = 0.5 * z, w = 1.0 - hz;
        return w + (((1.0 - w) - hz) + z * (z * pc));
    }
#endif
    double r = fma(n, -0x1.921fb54400000p+0, a);                   // pi/2, leading 33 bits (n * this is exact)
    r = fma(n, -0x1.0b4611a626331p-34, r);                         // pi/2, next 53 bits
    const double z = r * r;
    // sin r
    double ps = fma(z, 0x1.5d93a5acfd57cp-33, -0x1.ae5e68a2b9cebp-26);
    ps = fma(z, ps, 0x1.71de357b1fe7dp-19);
    ps = fma(z, ps, -0x1.a01a019c161d5p-13);
    ps = fma(z, ps, 0x1.111111110f8a6p-7);
    const double sn = fma(z * r, fma(z, ps, -0x1.5555555555549p-3), r);
    // cos r
    double pc = fma(z, -0x1.8fae9be8838d4p-37, 0x1.1ee9ebdb4b1c4p-29);
    pc = fma(z, pc, -0x1.27e4f809c52adp-22);
    pc = fma(z, pc, 0x1.a01a019cb1590p-16);
    pc = fma(z, pc, -0x1.6c16c16c15177p-10);
    pc = fma(z, pc, 0x1.555555555554cp-5);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double cs = w + (((1.0 - w) - hz) + z * (z * pc));
    const int q = (int)n & 3;
    const double v = (q & 1) ? sn : cs;
    return (q == 1 || q == 2) ? -v : v;
}

// Re w for regions II-IV (real part only).
__device__ __attribute__((noinline)) double faddeeva_re_core(double x, double y, double ax)
{
    const c64 z = {x, y};
    const double s = add_rn(ax, y);
    if (s > 5.5) {  // region II
        const c64 z2 = cmul(z, z);
        const c64 inner = {fma(z2.re, kInvSqrtPi, -1.4104739589), z2.im * kInvSqrtPi};
        const c64 n = cmul(z, inner);  // w = i n / d  ->  Re w = (n.re d.im - n.im d.re) / |d|^2
        c64 d = cmul(z2, c64{z2.re - 3.0, z2.im});
        d.re += 0.75;
        return fma(n.re, d.im, -(n.im * d.re)) * recip(fma(d.re, d.re, d.im * d.im));
    }
    const c64 t = {y, -x};
    // Regions III and IV are quotients of polynomials with REAL coefficients in a complex argument v = (a, b).  Such a
    // polynomial is evaluated by dividing it by the real quadratic that has v as a root, z^2 - r z + s with r = 2 a, s = |v|^2:
    //     b_n = c_n,  b_(n-1) = c_(n-1) + r b_n,  b_k = c_k + r b_(k+1) - s b_(k+2)   ->   p(v) = b_0 - conj(v) b_1
    // two real FMAs per coefficient where the complex Horner step of the reference (voigt.py:60-64, :70-84) takes four; the value
    // is the same polynomial's with a different rounding error: within 4e-15 of the exact value in region III; in region IV, where
    // the argument lies close to the real axis (small y) and the division loses a digit, Re w stays within 1.3e-13 of the formula's
    // exact value (the complex Horner form: 1.4e-14; 2e7 points against extended precision) — the line opacity of the full-size
    // workloads moved from 1.2e-14 to 3e-14 of the oracle's, tolerance 1e-12.  tests/test_gpu_hot_faddeeva.py pins the routine
    // point by point against the reference's vectors (2e-13).  The general Faddeeva evaluations are a fifth of the step with 1e6 lines
    // (profiles/r03_all_region1_experiment.txt).
    if (y >= sub_rn(mul_rn(0.195, ax), 0.176)) {  // region III: v = t = (y, -x)
        const double r = y + y, ms = -fma(x, x, y * y);
        double b2 = 0.5642236, b1 = fma(r, b2, 3.778987), b0;
        b0 = fma(r, b1, fma(ms, b2, 11.96482)), b2 = b1, b1 = b0;
        b0 = fma(r, b1, fma(ms, b2, 20.20933)), b2 = b1, b1 = b0;
        b0 = fma(r, b1, fma(ms, b2, 16.4955));
        const c64 p = {fma(-y, b1, b0), -(x * b1)};  // b0 - conj(v) b1, conj(v) = (y, x)
        double d1 = r + 6.699398, d2 = 1.0, d0;      // leading coefficient 1
        d0 = fma(r, d1, ms + 21.69274), d2 = d1, d1 = d0;
        d0 = fma(r, d1, fma(ms, d2, 39.27121)), d2 = d1, d1 = d0;
        d0 = fma(r, d1, fma(ms, d2, 38.82363)), d2 = d1, d1 = d0;
        d0 = fma(r, d1, fma(ms, d2, 16.4955));
        const c64 q = {fma(-y, d1, d0), -(x * d1)};
        return fma(p.re, q.re, p.im * q.im) * recip(fma(q.re, q.re, q.im * q.im));
    }
    // region IV: polynomials in v = -u (the nested "c - u (...)" form of the reference has all-positive coefficients in -u)
    c64 u = cmul(t, t);
    // exp(u.re) cos(u.im) first, and the polynomials only after it (the empty statement ties their inputs to its results): scheduled
    // the other way round the long cosine keeps the recurrences' operands alive and the routine needs two more registers than the
    // line kernel has to spare under its 72 (it spilled the scan's prefetched line index, 0.7 GB of scratch writes at 1e6 lines).
    // exp(u.re): -30.3 < u.re < 0.81 here; exp_neg is the device library's exp, operation for operation, on the argument -tau
    double ex = exp_neg(-u.re), cs = cos_small(u.im);
    asm("" : "+v"(ex), "+v"(cs), "+v"(u.re), "+v"(u.im));
    const double va = -u.re, vb = -u.im;
    const double r = va + va, ms = -fma(va, va, vb * vb);
    double b2 = 0.56419, b1 = fma(r, b2, 1.320522), b0;
    b0 = fma(r, b1, fma(ms, b2, 35.7668)), b2 = b1, b1 = b0;
    b0 = fma(r, b1, fma(ms, b2, 219.031)), b2 = b1, b1 = b0;
    b0 = fma(r, b1, fma(ms, b2, 1540.787)), b2 = b1, b1 = b0;
    b0 = fma(r, b1, fma(ms, b2, 3321.99)), b2 = b1, b1 = b0;
    b0 = fma(r, b1, fma(ms, b2, 36183.31));
    const c64 p = {fma(-va, b1, b0), vb * b1};  // b0 - conj(v) b1, conj(v) = (va, -vb)
    const c64 n = cmul(t, p);
    double d1 = r + 1.84144, d2 = 1.0, d0;
    d0 = fma(r, d1, ms + 61.5704), d2 = d1, d1 = d0;
    d0 = fma(r, d1, fma(ms, d2, 364.219)), d2 = d1, d1 = d0;
    d0 = fma(r, d1, fma(ms, d2, 2186.18)), d2 = d1, d1 = d0;
    d0 = fma(r, d1, fma(ms, d2, 9022.23)), d2 = d1, d1 = d0;
    d0 = fma(r, d1, fma(ms, d2, 24322.8)), d2 = d1, d1 = d0;
    d0 = fma(r, d1, fma(ms, d2, 32066.6));
    const c64 q = {fma(-va, d1, d0), vb * d1};
    const double frac = fma(n.re, q.re, n.im * q.im) * recip(fma(q.re, q.re, q.im * q.im));
    return fma(ex, cs, -frac);
}

// Per-(line, depth) constants the pre-pass stores for the line kernel.
//   inv_dw = 1 / doppler_width
//   y      = (gamma / (sqrt(pi) * pi)) / doppler_width        (voigt.py:148, exact operations)
//   amp    = alpha / (sqrt(pi) * doppler_width)               (voigt.py:149 and base.py:627)
// voigt_add returns acc + the term (see region1_add for why the sum is inside).
__device__ __forceinline__ double voigt_add_x(double acc, double x, double y, double amp, const RegionI& k)
{
    const double ax = fabs(x);
    if (add_rn(ax, y) > 15.0) return region1_add(acc, x, k);  // amp is inside k
    const double core = faddeeva_re_core(x, y, ax);
    asm("v_fma_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(amp), "v"(core));
    return acc;
}
__device__ __forceinline__ double voigt_add(double acc, double delta_nu, double inv_dw, double y, double amp, const RegionI& k)
{
    return voigt_add_x(acc, delta_nu * inv_dw, y, amp, k);
}
__device__ __forceinline__ double voigt_term(double delta_nu, double inv_dw, double y, double amp, const RegionI& k)
{
    return voigt_add(0.0, delta_nu, inv_dw, y, amp, k);
}

// ---- fp32 evaluation, mixed-precision mode only ---------------------------------------------------------------------
// The same four Humlicek regions evaluated in fp32 with the complex arithmetic PACKED: a complex number is one float2v
// (re, im), a complex Horner step p <- c + t p is two v_pk_fma_f32 (t.re * p + (c, 0), then (-t.im, t.im) * p.yx + that)
// where the fp64 routine spends four FMAs, exp and cos are the hardware's (v_exp_f32, v_cos_f32).  Measured against
// the fp64 routine over the golden (x, y) sets: tests/test_gpu_hot_faddeeva.py, tolerance 2e-5 of Re w (the W4
// approximation itself is good to 1e-4).
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2v pk_fma(float2v a, float2v b, float2v c) { return __builtin_elementwise_fma(a, b, c); }
// a * b for complex a, b given pm = (-a.im, a.im)
__device__ __forceinline__ float2v cmul32(float2v a, float2v pm, float2v b) { return pk_fma((float2v)(a.x), b, pm * b.yx); }
// (c, 0) + t p and (c, 0) - u p:  pm = (-t.im, t.im), resp. mp = (u.im, -u.im)
__device__ __forceinline__ float2v chorner_add32(float c, float tre, float2v pm, float2v p)
{
    return pk_fma(pm, p.yx, pk_fma((float2v)(tre), p, (float2v){c, 0.f}));
}
__device__ __forceinline__ float2v chorner_sub32(float c, float ure, float2v mp, float2v p)
{
    return pk_fma(mp, p.yx, pk_fma((float2v)(-ure), p, (float2v){c, 0.f}));
}
__device__ __forceinline__ float hsum32(float2v a) { return a.x + a.y; }
// acc + amp * Re w(x + i y), all regions
__device__ __forceinline__ float voigt_add32(float acc, float x, float y, float amp)
{
    const float ax = fabsf(x), s = ax + y;
    if (s > 15.f) {  // region I, the form of region1_add
        const float y2 = y * y;
        const float v = fmaf(x, x, y2 - 0.5f);
        const float yk = amp * (y * 0.5641895835f);
        return fmaf(fmaf(yk, v, yk), __builtin_amdgcn_rcpf(fmaf(v, v, y2 + y2)), acc);
    }
    float re;
    if (s > 5.5f) {  // region II
        const float2v z = {x, y}, zpm = {-y, y};
        const float2v z2 = cmul32(z, zpm, z);
        const float2v inner = pk_fma(z2, (float2v)(0.5641895835f), (float2v){-1.4104739589f, 0.f});
        const float2v n = cmul32(z, zpm, inner);
        const float2v d = pk_fma((float2v){-z2.y, z2.y}, (float2v){z2.y, z2.x - 3.0f}, pk_fma((float2v)(z2.x), (float2v){z2.x - 3.0f, z2.y}, (float2v){0.75f, 0.f}));
        const float2v cr = n * d.yx;  // (n.re d.im, n.im d.re)
        re = (cr.x - cr.y) * __builtin_amdgcn_rcpf(hsum32(d * d));
    } else {
        const float2v t = {y, -x}, tpm = {x, -x};  // (-t.im, t.im)
        if (y >= 0.195f * ax - 0.176f) {  // region III
            float2v p = pk_fma(t, (float2v)(0.5642236f), (float2v){3.778987f, 0.f});
            p = chorner_add32(11.96482f, y, tpm, p);
            p = chorner_add32(20.20933f, y, tpm, p);
            p = chorner_add32(16.4955f, y, tpm, p);
            float2v q = t + (float2v){6.699398f, 0.f};
            q = chorner_add32(21.69274f, y, tpm, q);
            q = chorner_add32(39.27121f, y, tpm, q);
            q = chorner_add32(38.82363f, y, tpm, q);
            q = chorner_add32(16.4955f, y, tpm, q);
            re = hsum32(p * q) * __builtin_amdgcn_rcpf(hsum32(q * q));
        } else {  // region IV
            const float2v u = cmul32(t, tpm, t), ump = {u.y, -u.y};
            float2v p = pk_fma(u, (float2v)(-0.56419f), (float2v){1.320522f, 0.f});
            p = chorner_sub32(35.7668f, u.x, ump, p);
            p = chorner_sub32(219.031f, u.x, ump, p);
            p = chorner_sub32(1540.787f, u.x, ump, p);
            p = chorner_sub32(3321.99f, u.x, ump, p);
            p = chorner_sub32(36183.31f, u.x, ump, p);
            const float2v n = cmul32(t, tpm, p);
            float2v q = (float2v){1.84144f, 0.f} - u;
            q = chorner_sub32(61.5704f, u.x, ump, q);
            q = chorner_sub32(364.219f, u.x, ump, q);
            q = chorner_sub32(2186.18f, u.x, ump, q);
            q = chorner_sub32(9022.23f, u.x, ump, q);
            q = chorner_sub32(24322.8f, u.x, ump, q);
            q = chorner_sub32(32066.6f, u.x, ump, q);
            const float frac = hsum32(n * q) * __builtin_amdgcn_rcpf(hsum32(q * q));
            // exp(u.re) cos(u.im): v_exp_f32 is 2^x, v_cos_f32 takes revolutions; -30.3 < u.re < 0.81, |u.im| < 10
            re = fmaf(__builtin_amdgcn_exp2f(u.x * 1.4426950409f), __builtin_amdgcn_cosf(u.y * 0.15915494309f), -frac);
        }
    }
    return fmaf(amp, re, acc);
}

// ---- radiative transfer pieces -------------------------------------------------------------
// radiation_field_solvers/base.py:22-45
__device__ __forceinline__ void rt_weights(double tau, double& w0, double& w1, double& w2)
{
    if (tau < 5e-4) {
        // tau/2 and tau/4 are exact scalings; tau/3 is a multiplication by the rounded 1/3 (|tau/3| < 2e-4 next to
        // 0.5: the last-place difference from a true division is below 1e-20 relative in w1)
        w0 = mul_rn(tau, sub_rn(1.0, mul_rn(tau, 0.5)));
        w1 = mul_rn(mul_rn(tau, tau), sub_rn(0.5, mul_rn(tau, 1.0 / 3)));
        w2 = mul_rn(mul_rn(mul_rn(tau, tau), tau), sub_rn(1.0 / 3, mul_rn(tau, 0.25)));
    } else if (tau < 50) {
        const double e = exp_neg(tau);
        w0 = sub_rn(1.0, e);
        w1 = sub_rn(w0, mul_rn(tau, e));
        w2 = sub_rn(mul_rn(2.0, w1), mul_rn(mul_rn(tau, tau), e));
    } else {
        w0 = 1.0;
        w1 = 1.0;
        w2 = 2.0;
    }
}
// blackbody.py:31-35
__device__ __forceinline__ double planck(double nu, double temp)
{
    const double pre = mul_rn(mul_rn(2.0, kH), mul_rn(mul_rn(nu, nu), nu)) / mul_rn(kC, kC);
    return pre / sub_rn(exp(mul_rn(kH, nu) / mul_rn(kKB, temp)), 1.0);
}
// The same function as the formal-solution kernels stage it, (N_d x N_nu) times per synthesis: the two IEEE divisions become
// refined reciprocals (each the correctly rounded quotient or its neighbour) and exp is exp_neg's sequence — h nu / (k T) lies
// in (0, 700) for anything a stellar atmosphere offers; outside that range, for T <= 0 or non-finite arguments the reference-order
// routine above runs instead.  Within 4 ulp of planck() (an ulp of the exponent x moves exp(x) by x ulp either way), 45 instructions
// instead of ~110.
__device__ __forceinline__ double planck_staged(double nu, double temp)
{
    const double kt = mul_rn(kKB, temp);
    const double x = mul_rn(kH, nu) * recip(kt);
    if (!(x > 1e-3 && x < 700.0 && kt > 1e-290 && kt < 1e290)) return planck(nu, temp);
    const double pre = mul_rn(mul_rn(2.0, kH), mul_rn(mul_rn(nu, nu), nu)) * (1.0 / (kC * kC));
    return pre * recip(exp_neg(-x) - 1.0);
}

// One short-characteristic step of the formal solution (radiation_field_solvers/base.py:200-266, van Noort 2002 eq. 14) as
// an AFFINE MAP of the incoming intensity, I[g+1] = c I[g] + e:
//     c = 1 - w0,   e = w0 S1 + w1 [(S1-S2) t0/t1 - (S1-S0) t1/t0] / (t0+t1) + w2 [(S2-S1)/t1 + (S0-S1)/t0] / (t0+t1)
// with t0, t1 the optical depths of this gap and the next, S0 S1 S2 the source at the three points.  Over the common denominator
// D = t0 t1 (t0 + t1) the two second-order terms are
//     [ (S0-S1) t1 (w1 t1 + w2) + (S2-S1) t0 (w2 - w1 t0) ] / D:
// ONE reciprocal per step and no 1/t0, 1/t1 (the kernels used to stage 1/mean-opacity and 1/ray-length tables for those).
// The weights (:22-45) keep the reference's operations and order; tau >= 50 needs no branch of its own: with tau clamped to 64
// the exponential form gives (1, 1, 2) exactly (e^-50 = 1.9e-22 is below half an ulp of 1, tau^2 e^-tau below half an ulp
// of 2).  Which lanes take the series (tau < 5e-4) is decided per lane, whether the block runs at all per wave.
// Anything unusual — t0 = 0 (no change, :203-206), t1 = 0, a denominator that is zero, subnormal, infinite or NaN — is caught
// by ONE class test of D and redone per lane in the reference's own form, IEEE divisions and all: the same inf / NaN pattern.
// LAST: the final gap (:253-266), e = w0 S1 + w2 (S0 - S1) / t0^2.
__device__ __forceinline__ void rt_weights_wave(double tau, double& w0, double& w1, double& w2)
{
    const bool small = tau < 5e-4;
    const unsigned long long m_small = __builtin_amdgcn_ballot_w64(small), m_all = __builtin_amdgcn_ballot_w64(true);
    // (no initial values: every lane that is read below has been written — the exponential form unless ALL lanes take the
    // series, the series in the lanes that select it; the empty statements only tell the compiler so.  Initial values, or an
    // early return for the all-series case, cost the common path three register copies per step)
    asm("; w0" : "=v"(w0));
    asm("; w1" : "=v"(w1));
    asm("; w2" : "=v"(w2));
    if (m_small != m_all) {
        // min(tau, 64) as the bare instruction (NaN -> 64: the reference's else-branch, (1, 1, 2), too); fmin() would first
        // canonicalise its argument with a v_max_f64
        double tc;
        asm("v_min_f64 %0, %1, %2" : "=v"(tc) : "v"(tau), "s"(64.0));
        const double e = exp_neg(tc);
        w0 = sub_rn(1.0, e);
        w1 = sub_rn(w0, mul_rn(tc, e));
        w2 = sub_rn(mul_rn(2.0, w1), mul_rn(mul_rn(tc, tc), e));
    }
    if (m_small) {
        const double a0 = mul_rn(tau, sub_rn(1.0, mul_rn(tau, 0.5)));
        const double t2 = mul_rn(tau, tau);
        const double a1 = mul_rn(t2, sub_rn(0.5, mul_rn(tau, 1.0 / 3)));
        const double a2 = mul_rn(mul_rn(t2, tau), sub_rn(1.0 / 3, mul_rn(tau, 0.25)));
        w0 = small ? a0 : w0;
        w1 = small ? a1 : w1;
        w2 = small ? a2 : w2;
    }
}
constexpr int kClassUnusual = 0x3FF & ~0x100;  // everything but a positive normal number: NaNs, infinities, zeros, subnormals, negatives
// the rare lanes, in the reference's own form (self-contained: nothing of the fast path has to stay alive for it)
template <bool LAST>
__device__ __forceinline__ void rt_coef_reference(double t0, double t1, double d10, double d21, double s1, double& c, double& e)
{
    if (t0 == 0.0) {  // :203-206, :253-254
        c = 1.0, e = 0.0;
        return;
    }
    double w0, w1, w2;
    rt_weights(t0, w0, w1, w2);
    c = sub_rn(1.0, w0);
    if constexpr (LAST) {
        e = add_rn(mul_rn(w0, s1), mul_rn(w2, d10) / mul_rn(t0, t0));
    } else {
        const double sum = add_rn(t0, t1);
        const double second = mul_rn(w1, sub_rn(mul_rn(-d21, t0 / t1), mul_rn(-d10, t1 / t0))) / sum;
        const double third = mul_rn(w2, add_rn(d21 / t1, d10 / t0)) / sum;
        e = add_rn(add_rn(mul_rn(w0, s1), second), third);
    }
}
// -> wave mask of the lanes whose (c, e) must be redone by rt_coef_reference
template <bool LAST>
__device__ __forceinline__ unsigned long long rt_coef_fast(double t0, double t1, double d10, double d21, double s1, double& c, double& e)
{
    double w0, w1, w2;
    rt_weights_wave(t0, w0, w1, w2);
    double den;
    if constexpr (LAST) {
        den = t0 * t0;
        e = fma(w0, s1, (w2 * d10) * recip(den));
    } else {
        den = (t0 * t1) * (t0 + t1);
        const double u = fma(w1, t1, w2), v = fma(-w1, t0, w2);
        const double num = fma(d10 * t1, u, (d21 * t0) * v);
        e = fma(w0, s1, num * recip(den));
    }
    c = 1.0 - w0;
    // (the wave mask straight from the compare: through a bool the compiler materialises 0 / 1 per lane and compares again)
    unsigned long long unusual;
    asm("v_cmp_class_f64 %0, %1, %2" : "=s"(unusual) : "v"(den), "s"(kClassUnusual));
    return unusual;
}
template <bool LAST>
__device__ __forceinline__ void rt_coef(double t0, double t1, double d10, double d21, double s1, double& c, double& e)
{
    const unsigned long long unusual = rt_coef_fast<LAST>(t0, t1, d10, d21, s1, c, e);
    if (unusual) {
        if ((unusual >> (threadIdx.x & 63)) & 1) rt_coef_reference<LAST>(t0, t1, d10, d21, s1, c, e);
    }
}

// ---- the formal solution's step in fp32: the tolerance path (mixed_precision = 1), stated tolerance 1e-4 on the flux -------------
// The same affine map I' = c I + e as rt_coef, from the EXACT weight functions (van Noort 2002 eq. 14)
//     w0 = 1 - e^-t,   w1 = 1 - e^-t (1 + t),   w2 = 2 - e^-t (2 + 2 t + t^2)
// instead of the reference's two-regime forms: below t = 0.25 their power series (six terms: truncation < 1e-7 relative), above
// it the exponential (v_exp_f32) — in fp32 the differences 1 - E ... lose 6e-8 / t of their value, which the series region
// keeps out of reach.  The reference's own forms differ from these functions by < 5e-8 (its series stops after two terms at
// t < 5e-4).  Optical depths below 1e-12 (no measurable change: the step adds t S) are treated as 0; in the second-order terms
// t is capped at 1e6 (they fall off as 1 / t: the cap moves them by < 1e-6 of the source difference) so that the common
// denominator t0 t1 (t0 + t1) stays inside the fp32 range.
__device__ __forceinline__ void rt_coef32(float t0, float t1, float d10, float d21, float s1, bool last, float& c, float& e)
{
    float w0, w1, w2;
    if (t0 < 0.25f) {
        w0 = t0 * fmaf(t0, fmaf(t0, fmaf(t0, fmaf(t0, fmaf(t0, -1.f / 720, 1.f / 120), -1.f / 24), 1.f / 6), -0.5f), 1.f);
        const float t2 = t0 * t0;
        w1 = t2 * fmaf(t0, fmaf(t0, fmaf(t0, fmaf(t0, fmaf(t0, -1.f / 840, 1.f / 144), -1.f / 30), 0.125f), -1.f / 3), 0.5f);
        w2 = (t2 * t0) * fmaf(t0, fmaf(t0, fmaf(t0, fmaf(t0, fmaf(t0, -1.f / 960, 1.f / 168), -1.f / 36), 0.1f), -0.25f), 1.f / 3);
    } else {
        const float E = __builtin_amdgcn_exp2f(fminf(t0, 100.f) * -1.4426950409f);
        w0 = 1.f - E;
        w1 = w0 - t0 * E;
        w2 = fmaf(-t0 * t0, E, w1 + w1);
    }
    c = 1.f - w0;
    const float a0 = fminf(t0, 1e6f), a1 = fminf(t1, 1e6f);
    if (last) {
        e = fmaf(w0, s1, (w2 * d10) * __builtin_amdgcn_rcpf(a0 * a0));
    } else {
        const float den = (a0 * a1) * (a0 + a1);
        const float u = fmaf(w1, a1, w2), v = fmaf(-w1, a0, w2);
        const float num = fmaf(d10 * a1, u, (d21 * a0) * v);
        e = fmaf(w0, s1, num * __builtin_amdgcn_rcpf(den));
        if (!(t1 >= 1e-12f)) e = w0 * s1;  // a transparent gap ahead: first order only (the reference divides by zero there)
    }
    if (!(t0 >= 1e-12f)) c = 1.f, e = 0.f;  // :203-206 (and NaN: no change rather than poison — tolerance path)
}
// The source function of one depth point AND its difference to the next one, S_d - S_{d+1}, for the fp32 formal solution.  The
// second-order terms multiply these differences by up to 1 / tau of a thin gap: formed from two rounded fp32 source values they
// carry 3e-7 S / |dS| — 2.9e-4 of the flux on a column whose temperature changes by 0.03 % across a gap ahead of a thin one
// (scripts/fuzz_raytrace.py, seed 849).  With x = h nu / k T:  S_d - S_{d+1} = S_d E expm1(x' - x) / (E' - 1),  x' - x =
// x (T - T') / T' with T - T' an exact fp64 difference: relative error ~1e-6 whatever the size of the difference.
__device__ __forceinline__ float expm1_32(float d)
{
    if (fabsf(d) < 0.25f)
        return d * fmaf(d, fmaf(d, fmaf(d, fmaf(d, fmaf(d, fmaf(d, 1.f / 5040, 1.f / 720), 1.f / 120), 1.f / 24), 1.f / 6), 0.5f), 1.f);
    return __builtin_amdgcn_exp2f(d * 1.4426950409f) - 1.f;
}
// hn = h nu, pre = 2 h nu^3 / c^2 (per frequency); inv_kt = 1 / (k T_d), inv_kt_next = 1 / (k T_{d+1}), rel = (T_d - T_{d+1}) / T_{d+1} (per depth)
__device__ __forceinline__ void planck32_pair(float hn, float pre, float inv_kt, float inv_kt_next, float rel, float& s, float& diff)
{
    const float x = hn * inv_kt, xn = hn * inv_kt_next;
    const float E = __builtin_amdgcn_exp2f(x * 1.4426950409f), En = __builtin_amdgcn_exp2f(xn * 1.4426950409f);
    s = pre * __builtin_amdgcn_rcpf(E - 1.f);
    const float rn = __builtin_amdgcn_rcpf(En - 1.f);
    diff = (E < 1e30f && En < 1e30f) ? (s * E) * (expm1_32(x * rel) * rn) : s - pre * rn;  // (beyond: both values are ~0)
}

}  // namespace sdx
