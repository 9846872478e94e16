// sdx_kernels.h — HIP kernels of the STARDIS hot path for gfx950.  Included once by stardis_hip.hip.
//
// Data layout in HBM (DESIGN.md section 3 has the table)
//   inputs   reference layout: line arrays [N_l][N_d] (doppler, alpha, gamma[N_l][N_d|1]), grid nus[N_nu] descending.
//   pre-pass WIDE items (half-width > 64 points) depth-major [N_d][N_l], split by use: scan words 16 B, records 48 B, slow part
//            16 B (+ 32 B fp32 records in the mixed mode); NARROW items line-major [N_l][N_d]: a one-byte half-width and three
//            f64 (1 / doppler, y, amplitude) — struct LineWork below.
//   outputs  [N_d][ld] with the frequency index contiguous.
#pragma once
#include <type_traits>

#include "sdx_math.h"
#include "sdx_broadening.h"
#include "sdx_cheb.h"

namespace sdx {

constexpr int kBlock = 256;

__device__ __forceinline__ void wave_sync()
{
    // LDS hand-over between lanes of ONE wave: the LDS unit executes a wave's instructions in order, so only the
    // compiler has to be kept from moving accesses across this point
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// ------------------------------------------------------------------------------------------------
// d_nu = -max(diff(nus))  (opacities_solvers/base.py:524-526): partial maxima, finished by consumers.
constexpr int kDnuPartials = 256;
// ... and, 4096 bytes behind the partial maxima, a SAMPLE of the grid: kGridSample frequencies at a stride of ceil(N_nu / kGridSample)
// points, written by the launch that computes the partial maxima and read — 16 contiguous KB — by every pre-pass block that
// looks for line centres (a block sampling the grid itself touches one 64-byte line per sample)
constexpr int kGridSample = 2048;
constexpr int kGridSampleOffset = 512;  // doubles
__device__ __forceinline__ void grid_sample_block(const int bid, const int n_blocks, int64_t n_nu, const double* __restrict__ nus, double* __restrict__ sample)
{
    const int64_t cstride = (n_nu + kGridSample - 1) / kGridSample;
    for (int q = bid * (int)blockDim.x + (int)threadIdx.x; q < kGridSample; q += n_blocks * (int)blockDim.x) {
        const int64_t j = (int64_t)q * cstride;
        sample[q] = j < n_nu ? nus[j] : -INFINITY;
    }
}

__device__ __forceinline__ void shard_range(int64_t n_nu, const double* __restrict__ nus, int64_t n_lines, const double* __restrict__ line_nus,
                                            int64_t nu_begin, int64_t nu_count, int* __restrict__ sel);

// ---- far field of the line opacity: the per-tile index ranges (the commentary is at far_eligible, below) ------------------------
constexpr double kFarRatio = 6.0;
constexpr int kFarTile = 256;  // the far field lives on the 256-point tiles of the fp64 line kernel (R = 4)
struct FarReq {
    int* range;            // [2 T], [2 T + 1] per GLOBAL tile T
    int64_t first, count;  // the tiles wanted (count = 0: none)
};
// centre and half-width of a tile from its end frequencies (exact halvings of one rounded sum / difference each)
__device__ __forceinline__ void far_tile_geometry(double nu_a, double nu_b, double& c, double& h)
{
    c = mul_rn(0.5, add_rn(nu_a, nu_b));
    h = mul_rn(0.5, sub_rn(nu_a, nu_b));
}
// first i in [lo, hi] with nus[i] < v (hi if none) on the descending grid
__device__ __forceinline__ int64_t first_below(const double* __restrict__ nus, int64_t lo, int64_t hi, double v)
{
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (nus[mid] >= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// Sixteen lanes per (tile, bound).  On a smooth grid the bound lies 2.5 tile widths beyond the tile's end, at an offset the tile's own
// mean spacing predicts to a few points: the lanes probe the sixteen grid points around that guess — ONE round trip behind the two
// loads of the tile's ends — and the count of frequencies >= the bound follows from their ballot.  Where the crossing is not among
// them (the spacing jumps), lane 0 searches: a bracket of eight tiles on that side first, then the rest of the grid.
__device__ __forceinline__ void far_ranges_block(const int bid, const int n_blocks, int64_t n_nu, const double* __restrict__ nus, const FarReq& fr)
{
    const int lane = threadIdx.x & 63, sub = lane & 15, grp_shift = lane & 48;
    const int64_t groups = (int64_t)n_blocks * (blockDim.x >> 4);
    for (int64_t k0 = (int64_t)bid * (blockDim.x >> 4); k0 < 2 * fr.count; k0 += groups) {  // (uniform trip count per block)
        const int64_t k = k0 + (threadIdx.x >> 4);
        const bool live = k < 2 * fr.count;
        const int64_t T = fr.first + (live ? k >> 1 : 0), i0 = T * kFarTile;
        const bool upper = (k & 1) == 0;
        int out = upper ? 0 : 0x7FFFFFFF;
        bool search = false;
        double v = 0.0;
        int64_t guess = 0;
        if (live && i0 + kFarTile <= n_nu) {
            double c, h;
            far_tile_geometry(nus[i0], nus[i0 + kFarTile - 1], c, h);
            if (h > 0.0) {
                search = true;
                v = upper ? add_rn(c, mul_rn(kFarRatio, h)) : sub_rn(c, mul_rn(kFarRatio, h));
                // (c +- 6 h lies 5 h beyond the tile's end: 5 / 2 * (kFarTile - 1) points at the tile's mean spacing)
                guess = upper ? i0 - (5 * (kFarTile - 1)) / 2 : i0 + kFarTile - 1 + (5 * (kFarTile - 1)) / 2;
            }
        }
        // the probes: grid points guess - 8 + sub (outside the grid: +inf above, -inf below, so that the ballot stays monotone)
        const int64_t pi = guess - 8 + sub;
        const double pv = !search ? 0.0 : (pi < 0 ? INFINITY : (pi >= n_nu ? -INFINITY : nus[pi]));
        const unsigned long long ge = __ballot(search && pv >= v);
        const unsigned m16 = (unsigned)(ge >> grp_shift) & 0xFFFFu;  // this group's sixteen answers (bit j: probe j >= v)
        if (search) {
            if ((m16 & 1u) && !(m16 & 0x8000u)) {
                out = (int)(guess - 8 + __popc(m16));  // = #{i : nus[i] >= v}: every point before the window is >= v as well
            } else if (sub == 0) {
                if (upper) {
                    const int64_t b0 = max(i0 - 8 * kFarTile, (int64_t)0);
                    out = (int)((b0 == 0 || nus[b0 - 1] >= v) ? first_below(nus, b0, i0, v) : first_below(nus, 0, b0 - 1, v));
                } else {
                    const int64_t b1 = min(i0 + 9 * kFarTile, n_nu);
                    out = (int)((b1 == n_nu || nus[b1] < v) ? first_below(nus, i0 + kFarTile, b1, v) : first_below(nus, b1 + 1, n_nu, v));
                }
            }
        }
        if (live && sub == 0) fr.range[2 * T + (upper ? 0 : 1)] = out;
    }
}
__global__ __launch_bounds__(kBlock) void k_far_ranges(int64_t n_nu, const double* __restrict__ nus, FarReq fr)
{
    far_ranges_block(blockIdx.x, gridDim.x, n_nu, nus, fr);
}


__global__ __launch_bounds__(kBlock) void k_dnu_partial(int64_t n_nu, const double* __restrict__ nus,
                                                        double* __restrict__ partial, int* __restrict__ zero, int64_t n_zero, FarReq far)
{
    far_ranges_block(blockIdx.x, gridDim.x, n_nu, nus, far);
    // (culled pre-pass: the per-line maxima the classification pass accumulates into are cleared here, not by a memset node)
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < n_zero; k += (int64_t)gridDim.x * kBlock) zero[k] = 0;
    grid_sample_block(blockIdx.x, gridDim.x, n_nu, nus, partial + kGridSampleOffset);
    double m = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i + 1 < n_nu; i += (int64_t)gridDim.x * kBlock)
        m = fmax(m, nus[i + 1] - nus[i]);
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    __shared__ double s[kBlock / 64];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) m = fmax(m, s[w]);
        partial[blockIdx.x] = m;
    }
}

__device__ __forceinline__ double block_dnu(const double* __restrict__ partial, int n_partial, double* s_red)
{
    double m = -INFINITY;
    for (int i = threadIdx.x; i < n_partial; i += blockDim.x) m = fmax(m, partial[i]);
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = s_red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmax(m, s_red[w]);
    __syncthreads();
    return -m;
}

// this thread's share of max(diff(nus)) — eight independent loads in flight per thread: the scan is latency-, not
// bandwidth-bound (the grid sits in L2).  Issued EARLY by the pre-pass so that it overlaps the centre search.
__device__ __forceinline__ double dnu_scan_local(const double* __restrict__ nus, int64_t n_nu)
{
    double m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
    const int64_t step = blockDim.x;
    int64_t i = threadIdx.x;
    for (; i + 3 * step + 1 < n_nu; i += 4 * step) {
        const double a0 = nus[i], b0 = nus[i + 1], a1 = nus[i + step], b1 = nus[i + step + 1];
        const double a2 = nus[i + 2 * step], b2 = nus[i + 2 * step + 1], a3 = nus[i + 3 * step], b3 = nus[i + 3 * step + 1];
        m0 = fmax(m0, b0 - a0);
        m1 = fmax(m1, b1 - a1);
        m2 = fmax(m2, b2 - a2);
        m3 = fmax(m3, b3 - a3);
    }
    for (; i + 1 < n_nu; i += step) m0 = fmax(m0, nus[i + 1] - nus[i]);
    return fmax(fmax(m0, m1), fmax(m2, m3));
}
__device__ __forceinline__ double block_max_to_dnu(double m, double* s_red)
{
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = s_red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmax(m, s_red[w]);
    __syncthreads();
    return -m;
}
__device__ __forceinline__ double block_dnu_scan(const double* __restrict__ nus, int64_t n_nu, double* s_red)
{
    return block_max_to_dnu(dnu_scan_local(nus, n_nu), s_red);
}

// index of the first grid frequency strictly below line_nu in the DESCENDING grid
//   = N_nu - searchsorted(nus[::-1], line_nu)   (base.py:556-558)
__device__ __forceinline__ double recip_guarded(double d)
{
    // Newton-refined hardware reciprocal for ordinary magnitudes, IEEE division otherwise (0, inf, NaN, subnormal)
    return (d > 1e-290 && d < 1e290) ? recip(d) : 1.0 / d;
}
constexpr double kInvSqrtPiPi = 0x1.6fcb5f827b97fp-3;  // 1 / (sqrt(pi) pi), correctly rounded
// 1 / dw, y and the amplitude of a (line, depth) item from its doppler width, gamma and alpha (voigt.py:148-149, base.py:627) — what the
// pre-pass stores in a narrow record, and what the narrow role forms itself when it reads the raw inputs (LineWork::narrow_raw)
__device__ __forceinline__ void narrow_params(double dw, double g, double a, double& inv, double& y, double& amp)
{
    inv = recip_guarded(dw);
    y = mul_rn(mul_rn(g, kInvSqrtPiPi), inv);
    amp = mul_rn(mul_rn(a, kInvSqrtPi), inv);
}
__device__ __forceinline__ int64_t closest_index(const double* __restrict__ nus, int64_t n_nu, double line_nu)
{
    int64_t lo = 0, hi = n_nu;  // first i with nus[i] < line_nu
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (nus[mid] >= line_nu) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// window rule base.py:561-575, bit-for-bit (each operation rounds once, same order)
// (the quotient by d_nu — one divisor for every item — through div_by: the correctly rounded quotient from 1 / d_nu, computed once)
__device__ __forceinline__ int64_t window_rule(int64_t c, int64_t n_nu, double d_nu, double r_dnu, double gamma, double dw, double alpha,
                                               int& lo, int& hi)
{
    const double pixels = mul_rn(div_by(mul_rn(add_rn(gamma, dw), alpha), d_nu, r_dnu), 20.0);
    const double forced = pixels > 10.0 ? pixels : 10.0;  // max(10, x); NaN keeps 10
    const int64_t hw = forced >= (double)n_nu ? n_nu : (int64_t)forced;  // int() truncates; saturating is equivalent
    const int64_t l = c - hw, h = c + hw;
    lo = (int)(l < 0 ? 0 : l);
    hi = (int)(h > n_nu ? n_nu : h);
    return hw;
}

// ------------------------------------------------------------------------------------------------
// Pre-pass: one block = kPreLines lines x up to 64 depths, 1024 threads, the (line, depth) items in registers (prepass_block).
// lines per pre-pass block — a template parameter: 16 (one item per thread) for short lists whose blocks all fit the chip at once,
// 32 (two items per thread: the block's latency chain is paid once for twice the items) for long ones, 48 / 54 (three) for the
// culled pre-pass of a frequency shard, whose blocks then fit ONE round of the chip
constexpr int kPreDepths = 64;
constexpr int kPreBlock = 1024;  // threads per pre-pass block: one (line, depth) item per thread, 16 waves to hide latency

constexpr int kNarrowHalfWidth = 64;    // windows with half-width <= this go to the narrow-window kernel
constexpr int kNarrowReach = 128;       // ... which looks for its lines within this many points of a frequency: line CORES up to this
                                        // half-width are delegated to it (lane <-> depth: one Faddeeva region per wave, where a
                                        // 64-point block of the wide role holds all four)
constexpr int kMediumHalfWidth = 4096;  // class bound of the indexed wide path: medium lines are found by centre range
constexpr int kMaxTile = 512;           // the widest tile of the wide role (64 R points, R <= 8)

// What the pre-pass leaves for the line kernels.  WIDE (line, depth) items (half-width > kNarrowHalfWidth) are depth-major
// [N_d][N_l] — a (depth, line-range) read is contiguous — and split by use:
//   wscan   16 B, read by every candidate test: window [lo, hi) and the CORE range [clo, chi), a superset of the grid points
//           where the Faddeeva region is not I (|x| + y <= 15); lo = hi = 0 for narrow / empty items
//   wrec    48 B, read once per (item, tile) that overlaps: line frequency, 1 / doppler and the region-I constants
//   wslow   16 B, read only by tiles that touch a window edge or the core: y and the amplitude (regions II-IV)
//   wrec32  32 B, the fp32 twin of wrec for the mixed-precision mode (line frequency as a hi + lo pair)
#ifdef SDX_WALK_STATS  // analysis build (scripts/r4/walk_stats.sh): what the waves of the line kernel spend their time on
constexpr int kWalkStatSlots = 1 << 19;  // lower half: wide waves by (depth < 64, tile < 512, subset < 8); upper half: narrow waves by frequency
__device__ unsigned long long g_walk_stats[(size_t)kWalkStatSlots * 8];
#endif
#ifdef SDX_PRE_STATS  // analysis build (scripts/r5/pre_stats.sh): when the phases of a pre-pass block begin (100 MHz device clock)
constexpr int kPreStatSlots = 1 << 14;
__device__ unsigned long long g_pre_stats[(size_t)kPreStatSlots * 8];
#define SDX_PRE_STAMP(n) do { if (tid == 0) pst[n] = wall_clock64(); } while (0)
#else
#define SDX_PRE_STAMP(n) do { } while (0)
#endif
struct alignas(16) WideScan {
    int lo, hi, clo, chi;
};
struct alignas(16) WideRec {
    double lnu, inv, yk, cv, cd, spare;
};
struct alignas(16) WideSlow {
    double y, amp;
};
struct alignas(32) WideRec32 {  // one s_load_dwordx8
    // ncl = -(lnu - nuh) * inv: the low part of the line frequency, already scaled.  Operands that meet in one instruction
    // share an aligned 8-byte pair (a VALU instruction reads one scalar register pair): (inv, ncl), (cv, cd), yk twice
    float nuh, yk, inv, ncl, cv, cd, pad0, pad1;
};
struct LineWork {
    WideScan* wscan;
    WideRec* wrec;
    WideSlow* wslow;
    WideRec32* wrec32;  // nullptr unless the mixed-precision mode is on
    // NARROW items (half-width <= kNarrowHalfWidth) and the delegated cores of wide items, LINE-major [N_l][N_d] so that
    // lane <-> depth reads coalesce.  Separate arrays: one 32-byte record per item was measured 45 % slower in the narrow
    // role (a wave's load then touches 28 cache lines three times over instead of 4 + 4 + 7 + 7 + 7)
    // half-width h of the narrow role's window [max(c - h, 0), min(c + h, N_nu)) around the line's centre c — a narrow item's
    // own half-width (<= kNarrowHalfWidth) or the delegated core's (<= kNarrowReach); 0: nothing to do for this (line, depth).
    // One byte instead of the two clamped bounds: the clamp is implied by the frequency index being a grid index.
    unsigned char* nhw;
    int skip_unlisted_scan;  // long lists, one depth block per line: wscan is written only for lines with a wide window somewhere
    double* n_inv;   // 1 / doppler
    double* n_y;
    double* n_amp;
    // narrow_raw != 0 (long dense lists, fp64; round 6): NO narrow records — n_inv / n_y / n_amp point at the CALLER'S doppler widths,
    // gammas and alphas (the same line-major layout; narrow_raw = the gammas' columns, 1 or N_d) and the narrow role forms 1 / dw, y and
    // the amplitude itself for the items it evaluates (narrow_params: the pre-pass's own three operations, hence the same bits): the
    // pre-pass writes one byte per narrow item instead of 25 — 1.35 of its 1.67 GB at 1e6 lines
    int narrow_raw;
    // mixed-precision mode: the narrow role evaluates in fp32 — the same three arrays as floats, and the line frequencies
    // as hi + lo float pairs [N_l]
    float* n_inv32;
    float* n_y32;
    float* n_amp32;
    float2v* lnu32;
    int* cnt_ge;     // [N_nu + 2]: number of lines whose centre index is >= p (lines are a prefix: centres descend)
    int* centre;     // [N_l] centre index of each line
    int* nhw_max;    // [N_l] largest NARROW half-width of the line over all depths (0: no narrow item)
    int* whw_max;    // [N_l] largest WIDE half-width of the line over all depths (0: no wide item)
    // long line lists: ascending indices of the lines whose widest window exceeds kMediumHalfWidth (they may reach any tile and
    // are scanned completely; all other lines are found by centre range).  nullptr for short lists (every line is scanned).
    int* hlist;
    int* hcount;     // [0] entries of hlist, [1] entries of wlist, [2] entries of xlist
    // frequency shards of long lists: the lines with a wide window whose centre lies within kMediumHalfWidth of the shard's
    // columns but outside the blocks of consecutive lines the shard prepares anyway (centres within 2 kNarrowReach) — with
    // hlist the lines the gather blocks of a culled pre-pass prepare.  nullptr otherwise.
    int* xlist;
    // ... and the other lines with a wide window (kNarrowHalfWidth < widest window <= kMediumHalfWidth), ascending, with
    // wrank[l] = number of wlist entries below line l ([N_l + 1]): the lines centred near a tile are a contiguous range of
    // line indices (cnt_ge), hence a contiguous range [wrank[la], wrank[lb]) of wlist
    int* wlist;
    int* wrank;
    WideScan* hscan;  // [N_d][N_l] rows, entry k of row d = wscan[d][hlist[k]]: the huge lines' scan words, contiguous
    // (list-ordered copies of the huge lines' RECORDS and of the wlist scan words as well were measured in round 3: the wide
    // role fetches 0.36 GB per launch at 1e6 lines either way — its 16 GB were the narrow role's — and the copies cost more
    // than the walk gained; removed)
    // frequency-sharded runs of long lists: the pre-pass only has to prepare the lines this shard can touch.  sel (device
    // memory, written on the side of k_dnu_partial; nullptr: every line): [0..1] = the index range [la, lb) of the lines whose
    // centre lies within kMediumHalfWidth of the shard's columns — the only ones a medium window can reach it from;
    // [2..3] = [na, nb), those within 2 kNarrowReach — every line there may touch the shard (narrow windows, delegated
    // cores) and is prepared by the range blocks; outside [na, nb) only lines with a wide window matter: hlist (any distance)
    // and xlist (within [la, lb)), prepared by the gather blocks
    const int* sel;
    int n_pix;      // pixel blocks (cnt_ge) behind the line blocks of the pre-pass grid
    int gather;     // gather blocks behind those: block g prepares hlist[g kPreLines ...] (0: none)
    int64_t shard_begin, shard_end;  // the shard's columns (culled runs only need cnt_ge near them)
    unsigned long long* evals;
    int* ticket;  // culled runs: the pre-pass launch's work counter, one per depth block (zeroed by k_hlist_count), or nullptr
    int front;    // culled runs: the blocks with work are the FIRST blocks of the pre-pass grid (k_line_prepass maps its block index)
    // far field (k_line_far): the (line, depth, tile) triples that far_eligible() accepts are left to k_line_far — the wide role
    // skips them — and their sum reaches the grid through the tile's 16 Chebyshev nodes (a third plane).
    // far_range[2 T], [2 T + 1] (k_far_ranges): a line is far from global tile T when its centre index is < the first or > the second
    // (0 and INT_MAX: tile T has no far field); nullptr = no far field in this launch
    const int* far_range;
};

// k / d for 0 <= k < 65536 and 1 <= d < 65536 with the divisor's reciprocal m = small_div_magic(d) = ceil(2^32 / d): one multiply-high
// instead of the ~30 instructions of a 32-bit division by a run-time value (the pre-pass indexes its (line, depth) items six times)
__device__ __forceinline__ unsigned small_div_magic(int d) { return d > 1 ? 0xFFFFFFFFu / (unsigned)d + 1u : 0u; }
__device__ __forceinline__ int small_div(int k, unsigned magic) { return magic ? (int)__umulhi((unsigned)k, magic) : k; }

template <bool GEN, int kPreLines>
__device__ __forceinline__ void prepass_block(const int bx, const int by, const int gy, int n_depth, int64_t n_nu, const double* __restrict__ nus,
                                                         const double* __restrict__ dnu_partial, int n_partial,
                                                         int64_t n_lines, const double* __restrict__ line_nus,
                                                         const double* __restrict__ doppler,
                                                         const double* __restrict__ gammas, int gamma_cols,
                                                         const double* __restrict__ alphas, LineWork w,
                                                         int* __restrict__ out_lo_ref, int* __restrict__ out_hi_ref,
                                                         int n_line_blocks, const LineParams& lp, const int tid = threadIdx.x)
{
    int gbx = -1;  // >= 0: a gather block
    if (bx >= n_line_blocks) {
        if (bx >= n_line_blocks + w.n_pix) {
            if (!w.gather) return;
            gbx = bx - n_line_blocks - w.n_pix;
        } else {
            // pixel blocks: cnt_ge[p] = #{l : centre_l >= p} = #{l : line_nu_l <= nus[p-1]}  (centre_l = #{i : nus[i] >= line_nu_l})
            if (by == 0 && w.cnt_ge) {
                const int64_t pidx = (int64_t)(bx - n_line_blocks) * blockDim.x + tid;
                // a shard only reads cnt_ge within kMediumHalfWidth (+ a narrow window) of the TILES that hold its columns: tiles
                // are aligned to the global grid, so the first and last one reach up to kMaxTile - 1 points beyond the shard
                const bool needed = !w.sel || (pidx >= w.shard_begin - kMediumHalfWidth - 2 * kNarrowReach - kMaxTile &&
                                               pidx <= w.shard_end + kMediumHalfWidth + 2 * kNarrowReach + kMaxTile);
                if (pidx <= n_nu + 1 && needed) {
                    int64_t cnt;
                    if (pidx == 0) cnt = n_lines;
                    else if (pidx == n_nu + 1) cnt = 0;
                    else {
                        const double v = nus[pidx - 1];
                        int64_t lo = 0, hi = n_lines;  // first l with line_nus[l] > v
                        while (lo < hi) {
                            const int64_t mid = lo + ((hi - lo) >> 1);
                            if (line_nus[mid] <= v) lo = mid + 1; else hi = mid;
                        }
                        cnt = lo;
                    }
                    w.cnt_ge[pidx] = (int)cnt;
                }
            }
            return;
        }
    }
    // Round 5: every thread keeps its (line, depth) items in REGISTERS, depth fastest — the order of the reference layout it reads
    // and of the line-major narrow arrays it writes: loads and narrow stores coalesce as they are, and nothing is staged through
    // LDS (rounds 1 - 4 transposed the block through 70 KB of it so that the depth-major wide records were written line-fastest:
    // three arrays written, read transposed, the derived constants written back and read again by a second pass, two more barriers).
    // The depth-major records of WIDE items — a few per cent of a list — and the scan words go out as scattered 16 / 48-byte stores;
    // the lines of a block are neighbours in those rows, so the pieces of a 64-byte sector meet in the L2 of the XCD the block runs
    // on before it is written back.  3 KB of LDS per block instead of 70: the size of a block is its threads and registers alone.
    // (54 lines per block: models of at most 56 depth points — every MARCS model has 56 — where 54 x 56 items are three per thread)
    constexpr int kItemDepths = kPreLines == 54 ? 56 : kPreDepths;
    constexpr int kPreItems = (kPreLines * kItemDepths + kPreBlock - 1) / kPreBlock;  // items per thread
    static_assert(kPreLines <= 64, "lines per pre-pass block");
#ifdef SDX_PRE_STATS
    unsigned long long pst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    SDX_PRE_STAMP(0);
    constexpr int kMaxWaves = kPreBlock / 64;
    __shared__ int64_t s_c[kPreLines];
    __shared__ double s_red[kMaxWaves];
    __shared__ unsigned long long s_ev[kMaxWaves];
    __shared__ int s_hwmax[kPreLines], s_whwmax[kPreLines];
    __shared__ double s_lnu[kPreLines];

    const int nthreads = blockDim.x;
    const bool gather = gbx >= 0;
    const int64_t l0 = (int64_t)(gather ? gbx : bx) * kPreLines;
    const int d0 = by * kPreDepths;
    // which lines this block prepares: kPreLines consecutive ones, or — gather blocks — kPreLines consecutive entries of hlist
    __shared__ int s_l[kPreLines];
    int nl = (int)min((int64_t)kPreLines, n_lines - l0);
    int n_h = 0;
    if (gather) {
        // the gather list: hlist, then xlist
        n_h = w.hcount[0];
        const int n_g = n_h + (w.xlist ? w.hcount[2] : 0);
        if (l0 >= n_g) return;  // block-uniform
        nl = min(kPreLines, n_g - (int)l0);
        if (tid < kPreLines) {
            const int k = (int)l0 + tid;
            s_l[tid] = tid < nl ? (k < n_h ? w.hlist[k] : w.xlist[k - n_h]) : 0;
        }
        __syncthreads();
    } else {
        if (w.sel && (l0 + kPreLines <= w.sel[2] || l0 >= w.sel[3])) return;  // not a block of the shard's own line range
        if (tid < kPreLines) s_l[tid] = (int)(l0 + tid);
        // (no barrier needed: the non-gather path indexes with l0 + ll directly)
    }
#define SDX_LINE_OF(ll) (gather ? (int64_t)s_l[ll] : l0 + (ll))
    const int nd = min(kPreDepths, n_depth - d0);
    const unsigned nd_magic = small_div_magic(nd);  // (items k < kPreLines kPreDepths = 4096)
    // line centres first: a sample of the grid in LDS brackets the answer — 128 points read from the grid itself where the block
    // also scans the grid spacing (<= 16384 points), otherwise the 2048 points the grid-spacing launch left (kGridSample): a
    // bracket of at most 64 points up to 131 072 grid points, i.e. no dependent global load before the one that resolves it
    // (rounds 1 - 4 sampled 128 points whatever the grid, and at 120 398 points a block spent 5.9 of its 26 us on four bisection
    // steps in global memory — device time stamps, scripts/r5/pre_stats.sh)
    __shared__ double s_coarse[kGridSample];
    const int n_samp = dnu_partial ? kGridSample : 128;  // (the launch that left the partial maxima also left the sample)
    const int64_t cstride = (n_nu + n_samp - 1) / n_samp;
    if (dnu_partial) {
        for (int q = tid; q < kGridSample; q += kPreBlock) s_coarse[q] = dnu_partial[kGridSampleOffset + q];
    } else if (tid < 128) {
        const int64_t j = (int64_t)tid * cstride;
        s_coarse[tid] = j < n_nu ? nus[j] : -INFINITY;
    }
    if (tid >= kPreBlock - kPreLines) {
        const int ll = tid - (kPreBlock - kPreLines);
        s_lnu[ll] = ll < nl ? line_nus[SDX_LINE_OF(ll)] : 0.0;
    }

    // this thread's share of max(diff(nus)), requested now — its loads travel while the centres are searched: one of the partial
    // maxima a launch before this one left (long grids: n_partial <= kDnuPartials < the block's threads), or its part of the scan
    const double dnu_local = dnu_partial ? (tid < n_partial ? dnu_partial[tid] : -INFINITY) : dnu_scan_local(nus, n_nu);
    __syncthreads();
    SDX_PRE_STAMP(1);  // the grid sample and the line frequencies are in LDS
    // The block's dense inputs are requested now (kPreItems per thread): their latency hides behind the centre search below —
    // whose one global load they precede in the queue — instead of following it; requested before the barrier above they only
    // held eighteen registers through the sample's hand-over.
    [[maybe_unused]] double r_dw[kPreItems], r_a[kPreItems], r_g[kPreItems];
    if constexpr (!GEN) {
#pragma unroll
        for (int it = 0; it < kPreItems; ++it) {
            const int k = tid + it * kPreBlock;
            r_dw[it] = r_a[it] = r_g[it] = 0.0;
            if (k < nl * nd) {
                const int ll = small_div(k, nd_magic), dd = k - ll * nd;
                const int64_t l = SDX_LINE_OF(ll);
                const int d = d0 + dd;
                r_dw[it] = doppler[l * n_depth + d];
                r_a[it] = alphas[l * n_depth + d];
                r_g[it] = gamma_cols > 1 ? gammas[l * gamma_cols + d] : gammas[l * gamma_cols];
            }
        }
    }
    // wave ll narrows the bracket of line ll to 64 points by bisection (none needed when the grid has <= 8192 points) and resolves
    // it with ONE coalesced load and a ballot — a chain of one or two dependent global loads instead of log2(N_nu / 128)
    for (int ll = tid >> 6; ll < nl; ll += kPreBlock / 64) {
        const int lane = tid & 63;
        {
            const double v = s_lnu[ll];
            int a = 0, b = n_samp;  // first sample strictly below v
            while (a < b) {
                const int mid = (a + b) >> 1;
                if (s_coarse[mid] >= v) a = mid + 1; else b = mid;
            }
            // the answer lies in ((a-1)*cstride, a*cstride]
            int64_t lo = a > 0 ? (int64_t)(a - 1) * cstride + 1 : 0;
            int64_t hi = min((int64_t)a * cstride, n_nu);
            while (hi - lo > 64) {
                const int64_t mid = lo + ((hi - lo) >> 1);
                if (nus[mid] >= v) lo = mid + 1; else hi = mid;
            }
            const int64_t idx = lo + lane;
            const bool below = idx < hi && nus[idx] < v;
            const unsigned long long m = __ballot(below);
            if (lane == 0) s_c[ll] = m ? lo + __builtin_ctzll(m) : hi;
        }
    }
    if (tid < kPreLines) s_hwmax[tid] = 0, s_whwmax[tid] = 0;
    SDX_PRE_STAMP(2);  // this wave's centres are found
    // d_nu (:524-526): from the partial maxima of k_dnu_partial, or — small grids — scanned here directly
    // (its barriers also publish the centres and the cleared maxima)
    // (block-uniform: both live in scalar registers — three items per thread leave the 64-VGPR budget no room for them)
    const double d_nu_v = block_max_to_dnu(dnu_local, s_red);
    const double d_nu = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(d_nu_v)), __builtin_amdgcn_readfirstlane(__double2loint(d_nu_v)));
    const double r_dnu_v = 1.0 / d_nu_v;
    const double r_dnu = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(r_dnu_v)), __builtin_amdgcn_readfirstlane(__double2loint(r_dnu_v)));
    SDX_PRE_STAMP(3);  // grid spacing known

    [[maybe_unused]] __shared__ GenDepth s_gd[GEN ? kPreDepths : 1];
    [[maybe_unused]] __shared__ GenLine s_gl[GEN ? kPreLines : 1];
    if constexpr (GEN) {
        // line parameters from per-line scalars and per-depth state (f1): nothing dense to read.  The per-depth and the
        // per-line factors (every pow / tgamma / n_eff) are evaluated once per block column / row and shared through LDS.
        if (tid < nd) s_gd[tid] = gen_depth(lp, d0 + tid);
        else if (tid >= 64 && tid < 64 + nl) s_gl[tid - 64] = gen_line(lp, line_nus[SDX_LINE_OF(tid - 64)], SDX_LINE_OF(tid - 64));
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kPreItems; ++it) {
            const int k = tid + it * kPreBlock;
            r_dw[it] = r_a[it] = r_g[it] = 0.0;
            if (k < nl * nd) {
                const int ll = small_div(k, nd_magic), dd = k - ll * nd;
                const GenDepth& D = s_gd[dd];  // fields are read from LDS where they are used: copies would cost ~40 VGPRs
                const GenLine& L = s_gl[ll];
                r_dw[it] = gen_doppler(lp, L, D);
                r_a[it] = gen_alpha(lp, L, D, d0 + dd, n_depth);
                r_g[it] = gen_gamma(lp, L, D);
            }
        }
    }

    // ONE arithmetic pass, depth fastest; every output of an item leaves from its thread
    constexpr bool kCountEvals = kPreLines < 48;  // (48 lines per block: culled shards only, which never count — two registers the block needs)
    [[maybe_unused]] unsigned long long ev = 0;
    // what the scan word of an item is rebuilt from after the barrier (two registers per item instead of the word's four: three
    // items per thread have to fit 64 VGPRs — two resident blocks per CU): the half-width of a WIDE window (0: the word is empty) and
    // the core's half-width, negative when the core is delegated to the narrow role
    int keep_hw[kPreItems], keep_chw[kPreItems];
#pragma unroll
    for (int it = 0; it < kPreItems; ++it) {
        const int k = tid + it * kPreBlock;
        keep_hw[it] = 0, keep_chw[it] = 0;
        if (k >= nl * nd) continue;
        const int ll = small_div(k, nd_magic), dd = k - ll * nd;
        const int64_t l = SDX_LINE_OF(ll);
        const double dw = r_dw[it], g = r_g[it], a = r_a[it];
        const int64_t c = s_c[ll];
        int lo, hi;
        const int64_t hw = window_rule(c, n_nu, d_nu, r_dnu, g, dw, a, lo, hi);
        const bool narrow = hw <= kNarrowHalfWidth;
        // 1 / dw once (the refined hardware reciprocal: the correctly rounded value but for ~1 in 1e8 arguments), y and the amplitude
        // as products with it — within 2 ulp of the reference's quotients (voigt.py:148-149), which the kernels multiply into x = dnu (1 / dw)
        // and the profile anyway; three divisions (~25 instructions each) fewer per item (round 6)
        double inv, yy, amp;
        narrow_params(dw, g, a, inv, yy, amp);
        int core_hw = 0;
        bool delegated = false;
        if (w.wscan) {
            const size_t o = (size_t)(d0 + dd) * n_lines + l;  // depth-major (wide kernel)
            WideScan sc = {0, 0, 0, 0};
            if (!narrow && hi > lo) {
                // core: grid points with |x| + y <= 15 lie within (15 - y) doppler widths of the line, i.e. within
                // floor(that / d_nu) + 1 points of the centre (d_nu is the SMALLEST spacing of the grid, :524-526).  A margin
                // (15.001, + 2 points) covers the rounding of x = delta_nu * (1 / dw); a superset costs nothing but speed.
                const double reach = mul_rn(mul_rn(15.001 - yy, dw), r_dnu);
                const int64_t chw = yy < 15.001 ? (reach >= (double)n_nu ? n_nu : (int64_t)reach + 2) : 0;
                keep_chw[it] = (int)min(chw, (int64_t)2147483647);
                sc.lo = lo;
                sc.hi = hi;
                sc.clo = max((int)max(c - chw, (int64_t)0), lo);
                sc.chi = chw > 0 ? min((int)min(c + chw, n_nu), hi) : sc.clo;
                // A core no wider than a narrow window is DELEGATED to the narrow role (lanes <-> depth: all depths of a line
                // sit in the same Faddeeva regions at one frequency, where a 64-point tile holds a few core points of one
                // depth): the narrow arrays below get the core's half-width as this item's window and the wide role leaves
                // those points out.
                delegated = chw > 0 && chw <= kNarrowReach && sc.chi > sc.clo;
                core_hw = (int)min(chw, hw);  // the core is clipped by the window (64 < hw < chw happens)
                if (delegated) {
                    atomicMax(&s_hwmax[ll], (int)chw);
                    sc.clo = -sc.clo - 1;  // the sign marks the delegation
                }
                const RegionI k1 = region1_setup(yy, amp);
                const double lnu = s_lnu[ll];
                w.wrec[o] = WideRec{lnu, inv, k1.yk, k1.cv, k1.cd, 0.0};
                w.wslow[o] = WideSlow{yy, amp};
                if (w.wrec32) {
                    const float nuh = (float)lnu;
                    w.wrec32[o] = WideRec32{nuh, (float)k1.yk, (float)inv, -((float)(lnu - (double)nuh) * (float)inv), (float)k1.cv, (float)k1.cd, 0.f, 0.f};
                }
                atomicMax(&s_whwmax[ll], (int)hw);
            } else if (narrow) {
                atomicMax(&s_hwmax[ll], (int)hw);
            }
            if (!narrow && hi > lo) keep_hw[it] = (int)hw, keep_chw[it] = delegated ? -keep_chw[it] : keep_chw[it];  // (the word is stored below, once the line's widest window is known)
            // a gather block knows its lines' positions in hlist: the list-ordered copy of the scan word is written here
            // (culled runs; otherwise k_hscan makes it once the list exists)
            if (gather && w.hscan && l0 + ll < n_h) w.hscan[(size_t)(d0 + dd) * n_lines + l0 + ll] = sc;
        }
        // line-major outputs, depth fastest: the stores coalesce
        const size_t o = (size_t)l * n_depth + (d0 + dd);
        if (out_lo_ref) {  // sdx_line_windows_dev (no wide records there: the window itself)
            out_lo_ref[o] = lo;
            out_hi_ref[o] = hi;
        }
        if (w.nhw) {
            // the narrow role's window: a narrow item's own half-width, the delegated core's, or nothing
            w.nhw[o] = (unsigned char)(narrow ? (int)hw : (delegated ? core_hw : 0));
            if ((narrow || delegated) && !w.narrow_raw) {
                if (w.n_inv32) {
                    w.n_inv32[o] = (float)inv;
                    w.n_y32[o] = (float)yy;
                    w.n_amp32[o] = (float)amp;
                } else {
                    w.n_inv[o] = inv;
                    w.n_y[o] = yy;
                    w.n_amp[o] = amp;
                }
            }
        }
        if constexpr (kCountEvals) {
            if (hi > lo) ev += (unsigned long long)(hi - lo);
        }
    }
    SDX_PRE_STAMP(4);  // arithmetic and per-item stores issued
    __syncthreads();
    SDX_PRE_STAMP(5);
    // the scan words (depth-major): every tile of the wide role scans the word of a line it considers — all lines of a short
    // list, the hlist / wlist lines of a long one.  A line of a long list without a wide window at any depth is in neither
    // list: its 16 bytes per depth (0.9 of 2.8 GB of pre-pass writes at 1e6 lines) are not written.
    if (w.wscan) {
#pragma unroll
        for (int it = 0; it < kPreItems; ++it) {
            const int k = tid + it * kPreBlock;
            if (k >= nl * nd) continue;
            const int ll = small_div(k, nd_magic), dd = k - ll * nd;
            if (w.skip_unlisted_scan && s_whwmax[ll] == 0) continue;
            WideScan sc = {0, 0, 0, 0};
            if (keep_hw[it] > 0) {  // the word of the arithmetic pass, from the same integers
                const int64_t c = s_c[ll], hw = keep_hw[it], chw = keep_chw[it] < 0 ? -(int64_t)keep_chw[it] : (int64_t)keep_chw[it];
                const int64_t lw = c - hw, hw_end = c + hw;
                sc.lo = (int)(lw < 0 ? 0 : lw);
                sc.hi = (int)(hw_end > n_nu ? n_nu : hw_end);
                sc.clo = max((int)max(c - chw, (int64_t)0), sc.lo);
                sc.chi = chw > 0 ? min((int)min(c + chw, n_nu), sc.hi) : sc.clo;
                if (keep_chw[it] < 0) sc.clo = -sc.clo - 1;
            }
            w.wscan[(size_t)(d0 + dd) * n_lines + SDX_LINE_OF(ll)] = sc;
        }
    }
    // per-line summary for the narrow kernel's candidate test: centre index and the largest narrow half-width
    if (w.nhw_max && tid < nl) {
        if (gy == 1) w.nhw_max[SDX_LINE_OF(tid)] = s_hwmax[tid], w.whw_max[SDX_LINE_OF(tid)] = s_whwmax[tid];
        else {  // deep models: several depth blocks per line, both arrays zeroed by the host first
            atomicMax(&w.nhw_max[SDX_LINE_OF(tid)], s_hwmax[tid]);
            atomicMax(&w.whw_max[SDX_LINE_OF(tid)], s_whwmax[tid]);
        }
        if (by == 0) w.centre[SDX_LINE_OF(tid)] = (int)s_c[tid];
        if (by == 0 && w.lnu32) {
            const double lnu = s_lnu[tid];
            w.lnu32[SDX_LINE_OF(tid)] = float2v{(float)lnu, (float)(lnu - (double)(float)lnu)};
        }
    }
    if constexpr (kCountEvals) {
        if (w.evals) {
            for (int off = 32; off > 0; off >>= 1) ev += __shfl_xor(ev, off);
            if ((tid & 63) == 0) s_ev[tid >> 6] = ev;
            __syncthreads();
            if (tid == 0) {
                for (int i = 1; i < (nthreads >> 6); ++i) ev += s_ev[i];
                if (ev) atomicAdd(w.evals, ev);
            }
        }
    }
#ifdef SDX_PRE_STATS
    if (tid == 0) {
        unsigned long long* const o = g_pre_stats + (size_t)((bx * 2 + (gather ? 1 : 0)) & (kPreStatSlots - 1)) * 8;
        o[0] = ((unsigned long long)bx << 8) | (gather ? 2 : 0) | 1;
        for (int q = 0; q < 6; ++q) o[1 + q] = pst[q];
        o[7] = wall_clock64();
    }
#endif
#undef SDX_LINE_OF
}

template <bool GEN, int LINES>
__global__ __launch_bounds__(kPreBlock) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8))) void k_line_prepass(int n_depth, int64_t n_nu, const double* __restrict__ nus,
                                                         const double* __restrict__ dnu_partial, int n_partial,
                                                         int64_t n_lines, const double* __restrict__ line_nus,
                                                         const double* __restrict__ doppler,
                                                         const double* __restrict__ gammas, int gamma_cols,
                                                         const double* __restrict__ alphas, LineWork w,
                                                         int* __restrict__ out_lo_ref, int* __restrict__ out_hi_ref,
                                                         int n_line_blocks, LineParams lp)
{
    // (dispatching the pixel and gather blocks of a culled shard BEFORE its line blocks — a rotated grid — was measured in round 4:
    // 47 against 44 us for this launch on an eighth of S-c3; the order stays [line | pixel | gather])
    int bx = blockIdx.x;
    if (w.sel && w.front) {
        // A culled shard's grid is one block per CANDIDATE (every block of consecutive lines twice over, 9 500 at S-c3) of which a
        // few hundred have work — somewhere in the middle of the range for most ranks, behind thousands of blocks that only look at
        // `sel` and leave, each holding a block slot (70 KB of LDS) for a round trip.  Here the blocks that have work are the FIRST of
        // the grid, whatever the rank: block index -> item by the counts the list launch left (range blocks, gather blocks, pixel
        // blocks; the long items first), and everything behind them returns.  Which block prepares an item does not matter.
        const int first = w.sel[2] / LINES, n_range = max((w.sel[3] + LINES - 1) / LINES - first, 0);
        const int n_g = (w.hcount[0] + (w.xlist ? w.hcount[2] : 0) + LINES - 1) / LINES;
        const int item = bx;
        if (item < n_range) bx = first + item;
        else if (item < n_range + n_g) bx = n_line_blocks + w.n_pix + (item - n_range);
        else if (item < n_range + n_g + w.n_pix) bx = n_line_blocks + (item - n_range - n_g);
        else return;
    }
    prepass_block<GEN, LINES>(bx, blockIdx.y, gridDim.y, n_depth, n_nu, nus, dnu_partial, n_partial, n_lines, line_nus, doppler, gammas,
                       gamma_cols, alphas, w, out_lo_ref, out_hi_ref, n_line_blocks, lp);
}

// The pre-pass of a CULLED shard.  Which blocks have work is only known on the device (sel, the list counts); one block per
// candidate is every block of the list twice over — 9 500 blocks for 600 with work at an eighth of S-c3, 62 000 for 5 000 at 1e6
// lines, each costing a dispatch, 70 KB of LDS and a dependent load before it returns (a third of that launch at 1e6 lines).  Here
// as many blocks as the chip holds draw the items that exist from a counter — the range blocks, then the gather blocks, then the
// pixel blocks — until none is left.  Which block prepares an item does not matter: every item writes its own outputs.  The thread index
// goes through an empty asm inside the loop: left visible, everything that depends only on it (a dozen addresses, the grid sample)
// is hoisted out of the loop and kept in registers — 91 VGPRs instead of 51, one resident block per CU instead of two.
template <int LINES>
__global__ __launch_bounds__(kPreBlock) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8))) void k_line_prepass_ticket(
    int n_depth, int64_t n_nu, const double* __restrict__ nus, const double* __restrict__ dnu_partial, int n_partial, int64_t n_lines,
    const double* __restrict__ line_nus, const double* __restrict__ doppler, const double* __restrict__ gammas, int gamma_cols,
    const double* __restrict__ alphas, LineWork w, int n_line_blocks, LineParams lp)
{
    __shared__ int s_item;
    const int first = w.sel[2] / LINES, n_range = max((w.sel[3] + LINES - 1) / LINES - first, 0);
    const int n_g = w.hcount[0] + (w.xlist ? w.hcount[2] : 0);
    const int total = w.n_pix + n_range + (n_g + LINES - 1) / LINES;
    for (;;) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if (tid == 0) s_item = atomicAdd(w.ticket + blockIdx.y, 1);
        __syncthreads();
        const int item = s_item;
        __syncthreads();  // (s_item is drawn again at the top; the block's LDS arrays serve its next item)
        if (item >= total) return;
        // (the long items first, the pixel blocks — a binary search per thread — last: they fill the launch's tail)
        const int n_long = total - w.n_pix;
        const int bx = item < n_range ? first + item : (item < n_long ? n_line_blocks + w.n_pix + (item - n_range) : n_line_blocks + (item - n_long));
        prepass_block<false, LINES>(bx, blockIdx.y, gridDim.y, n_depth, n_nu, nus, dnu_partial, n_partial, n_lines, line_nus, doppler, gammas, gamma_cols, alphas, w,
                                    nullptr, nullptr, n_line_blocks, lp, tid);
    }
}

// ------------------------------------------------------------------------------------------------
// Line opacity, wide windows, gather form.  One single-wave block owns (depth d, a tile of 64*R grid points, one
// of S line subsets); lane k owns grid points t0 + k + 64 r (r < R) and accumulates in registers.  The block
// streams its subset of the line list in chunks of 64 (chunk c belongs to subset c mod S): each lane tests one
// line's window against the tile, survivors are compacted (order-preserving, wave ballot) into LDS together with
// their depth-column constants, then every lane walks the compacted list.  Splitting the line list over S blocks
// shortens the serial chain of the deepest (hottest) layers, whose windows are widest; the S partial planes are
// added in subset order by the consumer (k_reduce_partials / k_total_alphas).  No atomics: bit-stable results.
// ------------------------------------------------------------------------------------------------
// Line opacity, wide windows, gather form.  One wave owns (depth d, a tile of 64 R grid points, one of S line subsets);
// lane k owns grid points t0 + k + 64 r (r < R) and accumulates in registers.  The wave goes through its candidate lines
// 64 at a time (chunk q of 64 consecutive candidates belongs to subset q mod S):
//   scan   each lane tests ONE candidate's window against the tile (16 B per candidate, coalesced) -> two wave masks:
//          `hit` (window overlaps the tile) and `fast` (the tile lies wholly inside the window and wholly outside the
//          line's core range, so every point is in Faddeeva region I and inside: no per-point tests);
//   walk   the hits in ascending order; the 48-byte record of a hit (line frequency, 1 / doppler, region-I constants) is
//          fetched from ONE address by the whole wave — scalar or broadcast loads, no LDS staging, no cross-lane traffic —
//          the next hit's record being requested before the current one is evaluated.
// Candidates: every line of the list (short lists), or — long lists — first the lines whose widest window exceeds
// kMediumHalfWidth (`hlist`, they may reach any tile) and then the lines whose CENTRE lies within kMediumHalfWidth of the
// tile, a contiguous index range read from cnt_ge (lines are sorted, so centres descend).  Chunks are counted from the start
// of the list (hlist position / line index), so the partition into subsets — and with it the summation order of a grid
// point — does not depend on the tile or on how the grid is sharded.
// The S subsets of a (depth, tile) are the S waves of one workgroup: their partial sums meet in LDS and wave 0 adds them in
// subset order and writes the tile — one line-opacity plane, no atomics, bit-stable results.
// ------------------------------------------------------------------------------------------------
// FAR FIELD.  The reference's window of a strong line spans thousands of grid points ((gamma + doppler) alpha / d_nu * 20 pixels,
// base.py:561-575): nine tenths of all window evaluations of the 3000 - 10000 A workloads lie more than two tiles away from
// the line's centre, where the profile is the region-I rational — a smooth function of frequency whose nearest singularity is
// the line itself.  For a full tile [i0, i1) of 64 R points with end frequencies nu_a > nu_b, centre c = (nu_a + nu_b) / 2 and
// half-width h = (nu_a - nu_b) / 2, a (line, depth) item is FAR when the tile lies wholly inside its window, clear of its core
// range (every point in region I) and |c - nu_l| >= 6 h (decided in index space, k_far_ranges).  The sum of the far items of a tile is then evaluated at the tile's
// kFarNodes = 16 Chebyshev nodes c + h cos(pi (j + 1/2) / 16) — the same rational, the same records, fp64 — and carried to
// the tile's grid points by the degree-15 interpolant (k_line_far): 16 evaluations per (item, tile) instead of 64 R.  The
// interpolation error of a function analytic inside the ellipse through a pole at distance D from the centre is
// ~ (D/h + sqrt((D/h)^2 - 1))^-16 <= 11.9^-16 = 6e-18 of the item's value; measured against 80-bit sums of the far items of
// S-c3's tiles: 2.5e-14 relative, rounding included (the direct fp64 sum: 4e-15).  The opacity tolerance is 1e-12, the flux's 1e-10.
// Which triples are far is a property of the grid and the list (global tiles), not of the shard or the launch geometry; both
// kernels decide it with THIS function on the same operands, so every triple is evaluated exactly once.
constexpr int kFarSplit = 8;  // line subsets (waves per workgroup) of k_line_far — the far field as a launch of its own (experiment knob SDX_FAR_LAUNCH, split-launch profiling): a shard's launch is a few hundred workgroups, its waves' chains are its duration
constexpr int kFarWaveLdsDoubles = 64 * 6 + 64;  // per wave of k_line_far: 64 staged records, the queue's line indices and tile masks
// The distance test in INDEX space, once per tile (one thread each): a line's centre index is cidx = #{i : nus[i] >= nu_l}
// (closest_index, the reference's own quantity), so with ihi = #{i : nus[i] >= c + 6 h} and ilo = #{i : nus[i] >= c - 6 h}
//   cidx < ihi  =>  nu_l > nus[ihi - 1] >= c + 6 h        cidx > ilo  =>  nu_l <= nus[ilo] < c - 6 h
// — rigorous on any descending grid, whatever its spacing does, and the kernels compare integers per candidate instead of
// carrying its frequency.  Tiles without a far field (cut by the grid's end, or of zero width) get (0, INT_MAX): never far.
// (far_ranges_block, near the top of this file: it rides in the launch that measures the grid's spacing.)
// The centre index itself is not needed: the core range of a scan word always holds it (prepass_block: clo <= c <= chi, both equal
// to c where the core is empty), so chi < ihi or clo > ilo decide the same thing from the 16 bytes every candidate test reads anyway
// — conservatively where a core is wide, which only leaves a few more tiles to the direct sum.
__device__ __forceinline__ bool far_eligible(const WideScan& sc, int i0, int i1, int ihi, int ilo)
{
    const int clo = sc.clo < 0 ? -sc.clo - 1 : sc.clo;
    return (sc.lo <= i0) & (sc.hi >= i1) & ((i1 <= clo) | (i0 >= sc.chi)) & ((sc.chi < ihi) | (clo > ilo));
}

constexpr int kWideLdsDoubles = 64 * 8;  // per wave: R <= 8 partial sums per lane
// ... and in the fp64 kernels with a far field, whose wide role queues its hits (line_wide_walk): 64 staged records (48 B), their
// scan words (16 B) and the queue itself (64 ints)
constexpr int kWideFarLdsDoubles = 64 * 6 + 64 * 2 + 32;
#ifndef SDX_WIDE_QUEUED  // (A/B builds: 0 = the direct walk in the far-field kernels too)
#define SDX_WIDE_QUEUED 1
#endif
#ifndef SDX_SCAN_BATCH   // (A/B builds: chunks of candidates requested together by a queued walk)
#define SDX_SCAN_BATCH 1
#endif
#ifndef SDX_FAR_WAVES    // (A/B builds: waves per SIMD the fp64 far-field line kernels are compiled for)
#define SDX_FAR_WAVES 6
#endif

template <int R, int STRIDE = kWideLdsDoubles>
__device__ __forceinline__ void wide_reduce_and_store(const int split, const int n_split, double (&acc)[R], const int idx0, const int64_t s0,
                                                      const int64_t s1, double* __restrict__ lds_all, double* __restrict__ plane, int64_t pld,
                                                      const int d)
{
    const int lane = threadIdx.x & 63;
    if (n_split > 1) {
        if (split > 0) {
            double* mine = lds_all + (size_t)split * STRIDE;
#pragma unroll
            for (int r = 0; r < R; ++r) mine[r * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (split == 0) {
#pragma unroll 1  // (unrolled eight times the reads of all subsets are hoisted above the sums and spill)
            for (int s = 1; s < n_split; ++s) {
                const double* other = lds_all + (size_t)s * STRIDE;
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = add_rn(acc[r], other[r * 64 + lane]);
            }
        }
    }
    if (split == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (idx0 + 64 * r >= s0 && idx0 + 64 * r < s1) plane[(size_t)d * pld + (idx0 + 64 * r - s0)] = acc[r];  // the shard's columns only
    }
}

// fp32 evaluation of the region-I rational for the mixed-precision mode, TWO grid points per instruction (v_pk_add / mul /
// fma_f32: a packed instruction issues like one fp64 instruction and does two evaluations; only the reciprocal is per
// point).  x = fma(dq, inv, c0) like the fp64 walk's: dq = the lane's frequencies as OFFSETS from the tile's base frequency
// (formed in fp64, rounded once: 1.2e4 Hz at the far end of a tile, 1e-5 of a Doppler width) and c0 = (base - nu_l) * inv from
// the hi / lo split of the line frequency, formed once per (line, tile) — ONE packed instruction per pair of points where rounds
// 2 - 4 spent three on the hi / lo split of both frequencies (round 5: 18 -> 14 packed instructions per line and lane).
// Everything else is plain fp32 (v_rcp_f32 is good to 1 ulp).
__device__ __forceinline__ float region1_c0(float base_h, const WideRec32& k)
{
    return fmaf(base_h - k.nuh, k.inv, k.ncl);  // ((base_h - nuh) - nul) * inv: the difference of two fp32 frequencies is exact
}
__device__ __forceinline__ float2v region1_f32x2(float2v acc, float2v dq, float c0, const WideRec32& k)
{
    const float2v x = __builtin_elementwise_fma(dq, (float2v)(k.inv), (float2v)(c0));
    const float2v v = __builtin_elementwise_fma(x, x, (float2v)(k.cv));
    const float2v den = __builtin_elementwise_fma(v, v, (float2v)(k.cd));
    const float2v num = __builtin_elementwise_fma((float2v)(k.yk), v, (float2v)(k.yk));
    float2v r;
    r.x = __builtin_amdgcn_rcpf(den.x);
    r.y = __builtin_amdgcn_rcpf(den.y);
    // no inline asm here: its operands come straight from v_rcp_f32, and the compiler does not insert the wait state a
    // transcendental result needs before a VALU use when the user is an asm statement
    return __builtin_elementwise_fma(num, r, acc);
}

// DEFER: the partial sums of this wave are handed back (acc_out) instead of being reduced and stored here — the kernel of dense
// long lists has ONE reduction for both roles (line_all_body)
// STAGED (fp32-mixed mode, the kernel of very dense lists): the records of a chunk's hits reach the lanes through LDS (below)
template <int R, bool MIXED, bool DEFER = false, bool STAGED = false, bool FAR = false>
__device__ __forceinline__ void line_wide_walk(const int tile_idx, const int split, const int n_split, const int d, int64_t n_nu,
                                               const double* __restrict__ nus, int64_t nu_begin, int64_t nu_count, int64_t n_lines,
                                               LineWork w, double* __restrict__ plane, int64_t pld, double* __restrict__ lds_all,
                                               double* __restrict__ acc_out = nullptr)
{
    constexpr int kTile = 64 * R;
    // GLOBAL tiles: tile boundaries are multiples of kTile from grid index 0 whatever the shard, and a tile cut by a shard
    // boundary is classified and evaluated whole (only the stores are masked).  How a (line, depth, tile) is treated — test-free,
    // edge, core — and which four points of a lane share one reciprocal is then a property of the grid alone: a frequency shard
    // produces the bits of the unsharded run.
    const int64_t t0 = (nu_begin / kTile + (int64_t)tile_idx) * kTile;
    const int64_t t1 = min(t0 + kTile, n_nu);
    const int64_t s0 = nu_begin, s1 = nu_begin + nu_count;  // columns stored by this launch
    const int lane = threadIdx.x & 63;
    const int it0 = (int)t0, it1 = (int)t1;

    // fp64 mode: the lane's frequencies as offsets from the tile's first frequency (exact: neighbouring grid frequencies are
    // within a factor of two), so that x = (nu_i - nu_l) / doppler is ONE instruction per point,
    // x = fma(dnu_i, inv, c0) with c0 = (nu_base - nu_l) * inv formed once per (line, tile).  Mixed mode: hi + lo fp32 pairs only
    // (the fp64 value is their exact sum to 2^-48, rebuilt on the rare general path), so that R = 8 points per lane fit the
    // register budget
    double dnu[MIXED ? 1 : R], acc[R];
    float2v dq[MIXED ? R / 2 : 1], acc32[MIXED ? R / 2 : 1];  // pairs of points (r, r + 1): offsets from the tile's base frequency, fp32 sums
    // (grid index of point r of this lane: it0 + lane + 64 r, formed where it is needed — an edge test, the final store —
    // instead of living in registers through the walk)
    const int idx0 = it0 + lane;
    // (both halves through readfirstlane: the base is wave-uniform and ends up in an SGPR pair)
    const double nu_base_v = nus[t0];
    const double nu_base = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(nu_base_v)), __builtin_amdgcn_readfirstlane(__double2loint(nu_base_v)));
    // ... and a copy pinned in a VGPR pair: c0 = (nu_base - nu_l) * inv takes the record's nu_l as its scalar operand (a VALU
    // instruction reads one SGPR pair), otherwise the compiler moves nu_l into vector registers for every line it walks
    double nu_base_vec = nu_base;
    asm("" : "+v"(nu_base_vec));
    // mixed mode: the base frequency as an fp32 number (in a VGPR: it meets the record's scalar operands in one instruction)
    float base_h = (float)nu_base;
    asm("" : "+v"(base_h));
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = t0 + lane + r * 64;
        const double nu = i < t1 ? nus[i] : nu_base;
        acc[r] = 0.0;
        if constexpr (MIXED) {
            dq[r >> 1][r & 1] = (float)(nu - (double)base_h);  // (exact in fp64, rounded once)
            acc32[r >> 1][r & 1] = 0.f;
        } else {
            dnu[r] = nu - nu_base;
        }
    }
    // far field: the triples k_line_far evaluates at the tile's Chebyshev nodes are no hits here (full tiles only)
    // (FAR is a kernel of its own: the candidate's centre index is one more register carried through the walk, and the walk's
    // budget has none to spare — with the test in the one kernel its sums spilled to scratch)
    static_assert(!FAR || kTile == kFarTile, "the far field lives on 256-point tiles");
    [[maybe_unused]] int far_ihi = 0, far_ilo = 0x7FFFFFFF;  // (never far)
    if constexpr (FAR) {
        far_ihi = __builtin_amdgcn_readfirstlane(w.far_range[2 * (t0 / kTile)]);
        far_ilo = __builtin_amdgcn_readfirstlane(w.far_range[2 * (t0 / kTile) + 1]);
    }
    const size_t row = (size_t)d * n_lines;
    const WideScan* __restrict__ scan_row = w.wscan + row;
    const WideScan* __restrict__ hscan_row = w.hscan + row;
    const WideRec* __restrict__ rec_row = w.wrec + row;
    const WideSlow* __restrict__ slow_row = w.wslow + row;
    const WideRec32* __restrict__ rec32_row = MIXED ? w.wrec32 + row : nullptr;
    const int n_h = w.hlist ? __builtin_amdgcn_readfirstlane(*w.hcount) : 0;
    int pending32 = 0;  // fp32 terms accumulated since the last flush into the fp64 sums
#ifdef SDX_WALK_STATS
    const unsigned long long st_t0 = wall_clock64();
    int st_chunks = 0, st_fast = 0, st_general = 0, st_if = 0, st_slow = 0, st_flush = 0;  // (st_flush: ticks spent evaluating queued hits)
#endif

    // With a far field the hits that remain are few (the near zone of a line, window edges, cores: ~1 per chunk of 64 candidates
    // at S-c3 where the direct walk met 14), and a wave that fetches the record of every hit where it meets it is one dependent
    // round trip after the other (measured: 3.4 us per chunk).  The fp64 kernels with a far field therefore QUEUE their hits in LDS —
    // line index + test-free flag, list order — and evaluate a full queue (64 hits) at once: lane j fetches hit j's record and scan
    // word, all in one round trip, into LDS; the wave then goes through them with broadcast reads.  Same hits, same order, same
    // arithmetic as the direct walk.
    constexpr bool QUEUED = FAR && !MIXED && SDX_WIDE_QUEUED;
    constexpr int kLdsStride = QUEUED ? kWideFarLdsDoubles : kWideLdsDoubles;
    [[maybe_unused]] WideRec* const q_rec = reinterpret_cast<WideRec*>(lds_all + (size_t)split * kLdsStride);  // 64 x 48 B
    [[maybe_unused]] WideScan* const q_sc = reinterpret_cast<WideScan*>(lds_all + (size_t)split * kLdsStride + 64 * 6);
    [[maybe_unused]] int* const q_ent = reinterpret_cast<int*>(lds_all + (size_t)split * kLdsStride + 64 * 8);
    [[maybe_unused]] int q_count = 0;
    [[maybe_unused]] auto q_flush = [&](int cnt) {
      if constexpr (QUEUED) {
        if (cnt == 0) return;
        wave_sync();
        const bool mine = lane < cnt;
        const int ent = mine ? q_ent[lane] : 0;
        const int ql = ent & 0x7FFFFFFF;
        if (mine) {
            q_rec[lane] = rec_row[ql];
            q_sc[lane] = scan_row[ql];
        }
        const unsigned long long mfq = __ballot(mine && ent < 0);
        wave_sync();
        for (int k = 0; k < cnt; ++k) {
            const WideRec cur = q_rec[k];  // (one address for the whole wave: a broadcast read)
            const RegionI k1 = {cur.yk, cur.cv, cur.cd};
            const double c0 = (nu_base_vec - cur.lnu) * cur.inv;
            if ((mfq >> k) & 1) {
                region1_add_shared<R>(acc, dnu, cur.inv, c0, k1);
#ifdef SDX_WALK_STATS
                ++st_fast;
#endif
                continue;
            }
#ifdef SDX_WALK_STATS
            ++st_general;
#endif
            // the tile touches a window edge or the core (the direct walk below has the commentary)
            const WideScan hs = q_sc[k];
            const int jlo = __builtin_amdgcn_readfirstlane(hs.lo), jhi = __builtin_amdgcn_readfirstlane(hs.hi);
            const int jc = __builtin_amdgcn_readfirstlane(hs.clo), jchi = __builtin_amdgcn_readfirstlane(hs.chi);
            const bool delegated = jc < 0;
            const int jclo = delegated ? -jc - 1 : jc;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int a = it0 + 64 * r, z = min(a + 64, it1);
                if (z <= jlo || a >= jhi || a >= it1) continue;
                const bool over_core = !(z <= jclo || a >= jchi);
#ifdef SDX_WALK_STATS
                if (!over_core || delegated) ++st_if; else ++st_slow;
#endif
                if (!over_core || delegated) {
                    const int ir = idx0 + 64 * r;
                    const bool take = ir >= jlo && ir < jhi && !(over_core && ir >= jclo && ir < jchi);
                    acc[r] = region1_add_if(acc[r], fma(dnu[MIXED ? 0 : r], cur.inv, c0), k1, take);
                } else {
                    const WideSlow sl = slow_row[__builtin_amdgcn_readlane(ql, k)];
                    if (idx0 + 64 * r >= jlo && idx0 + 64 * r < jhi) acc[r] = voigt_add_x(acc[r], fma(dnu[MIXED ? 0 : r], cur.inv, c0), sl.y, sl.amp, k1);
                }
            }
        }
        wave_sync();  // (the queue and the staged records are written again)
      }
    };

    for (int pass = w.hlist ? 0 : 1; pass < 2; ++pass) {
        // candidate positions [ka, kb) of this pass: hlist positions (pass 0); wlist positions or — short lists — line indices (pass 1)
        int ka = 0, kb = n_h;
        if (pass == 1) {
            kb = (int)n_lines;
            if (w.hlist) {  // lines whose centre c satisfies t0 - H < c < t1 + H: line indices [la, lb), wlist positions [wrank[la], wrank[lb])
                const int64_t pa = max(t0 - kMediumHalfWidth + 1, (int64_t)0), pb = min(t1 + kMediumHalfWidth - 1, n_nu);
                ka = __builtin_amdgcn_readfirstlane(w.wrank[w.cnt_ge[pb + 1]]);
                kb = __builtin_amdgcn_readfirstlane(w.wrank[w.cnt_ge[pa]]);
                // A culled shard has prepared only the wlist lines centred within kMediumHalfWidth of ITS OWN columns (range
                // blocks + xlist); a tile cut by the shard boundary reaches up to kTile - 1 points further, and the records of the
                // lines centred out there are whatever the workspace held.  None of them can reach a STORED column (half-width <=
                // kMediumHalfWidth), and a candidate's subset is its position in wlist, so leaving them out changes no stored bit.
                if (w.sel) {
                    ka = max(ka, __builtin_amdgcn_readfirstlane(w.wrank[w.sel[0]]));
                    kb = min(kb, __builtin_amdgcn_readfirstlane(w.wrank[w.sel[1]]));
                }
            }
        }
        if (kb <= ka) continue;
        const int q_first = ka >> 6, q_last = (kb - 1) >> 6;
        int q = q_first + ((split - q_first % n_split) + n_split) % n_split;  // first chunk >= q_first of this subset
        // the scan word of the NEXT chunk is requested behind the current chunk's arithmetic
        auto fetch = [&](int qq, int& line, WideScan& sc) {
            const int k = qq * 64 + lane;
            line = -1;
            sc = WideScan{0, 0, 0, 0};
            if (qq <= q_last && k >= ka && k < kb) {
                if (pass == 0) {
                    line = w.hlist[k];
                    sc = w.hscan ? hscan_row[k] : scan_row[line];
                } else {
                    line = w.hlist ? w.wlist[k] : k;
                    sc = scan_row[line];
                }
            }
        };
        if constexpr (QUEUED) {
            // the scan of a queued walk has nothing to wait for but its own loads: kScanBatch chunks are requested together (a chunk is
            // two dependent loads in the wlist pass), tested, and their hits appended in list order
            constexpr int kScanBatch = SDX_SCAN_BATCH;
            for (; q <= q_last; q += kScanBatch * n_split) {
                int bl[kScanBatch];
                WideScan bs[kScanBatch];
#pragma unroll
                for (int u = 0; u < kScanBatch; ++u) fetch(q + u * n_split, bl[u], bs[u]);
#pragma unroll
                for (int u = 0; u < kScanBatch; ++u) {
                    const int line = bl[u];
                    const WideScan sc = bs[u];
                    const bool far = far_eligible(sc, it0, it1, far_ihi, far_ilo);  // k_line_far's
                    const bool hit = (line >= 0) & (sc.lo < it1) & (sc.hi > it0) & !far;
                    const int clo = sc.clo < 0 ? -sc.clo - 1 : sc.clo;
                    const bool fast = hit & (sc.lo <= it0) & (sc.hi >= it1) & ((it1 <= clo) | (it0 >= sc.chi));
                    const unsigned long long m = __ballot(hit);
#ifdef SDX_WALK_STATS
                    if (q + u * n_split <= q_last) ++st_chunks;
#endif
                    if (m == 0) continue;
                    const int n = __popcll(m);
                    if (q_count + n > 64) {
#ifdef SDX_WALK_STATS
                        const unsigned long long tq = wall_clock64();
#endif
                        q_flush(q_count);
#ifdef SDX_WALK_STATS
                        st_flush += (int)(wall_clock64() - tq);
#endif
                        q_count = 0;
                    }
                    if (hit) q_ent[q_count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = line | (fast ? (int)0x80000000 : 0);
                    q_count += n;
                }
            }
            continue;
        }
        int line_next;
        WideScan sc_next;
        fetch(q, line_next, sc_next);
        for (; q <= q_last; q += n_split) {
            const int line = line_next;
            const WideScan sc = sc_next;
            fetch(q + n_split, line_next, sc_next);
            const bool far = FAR && far_eligible(sc, it0, it1, far_ihi, far_ilo);  // k_line_far's
            const bool hit = (line >= 0) & (sc.lo < it1) & (sc.hi > it0) & !far;  // narrow / empty items have lo = hi = 0
            const int clo = sc.clo < 0 ? -sc.clo - 1 : sc.clo;                // sign: core delegated to the narrow role
            const bool fast = hit & (sc.lo <= it0) & (sc.hi >= it1) & ((it1 <= clo) | (it0 >= sc.chi));
            unsigned long long m = __ballot(hit);
#ifdef SDX_WALK_STATS
            ++st_chunks;
#endif
            if (m == 0) continue;
            const unsigned long long mf = __ballot(fast);
            if constexpr (MIXED) pending32 += __popcll(m);
            if constexpr (MIXED && STAGED) {
                // fp32-mixed mode, lists whose tiles find most of a chunk's candidates hitting (1e6 lines: 10 000 lines wider than
                // 4096 points, every tile walks them all): the hits of the chunk are COMPACTED — one wave permutation puts hit j's line index, lane and
                // test-free flag into lane j — and lane j fetches hit j's 32-byte fp32 record and stages it in LDS.  The walk is
                // then over hit ordinals: a run of test-free hits is a counted loop whose records arrive by broadcast LDS reads,
                // two per trip — no mask arithmetic, no address arithmetic, no scalar load per hit.  (Round 4's walk fetched every
                // record with scalar loads: ~20 scalar instructions per hit — lowest set bit, read-lane, 64-bit address, two
                // s_load, the test for the next hit — next to ~19 vector ones.)  Same records, same order, same pairing of the fp32
                // sums: the bits of the scalar walk.  Measured in round 5 (two builds, alternating): the line kernel of the 1e6-line
                // list 7.59 -> 6.70 ms, that of the 1.5e5-line list 1.60 -> 1.69 ms (a few hits per chunk: the staging of a chunk —
                // three permutations, a gathered load, two hand-overs through LDS — costs more than its scalar fetches) — hence only
                // in the kernel that very dense lists run.
                WideRec32* const stage = reinterpret_cast<WideRec32*>(lds_all + (size_t)split * kWideLdsDoubles);  // 64 x 32 B of this wave's LDS
                const int n_hits = __popcll(m);
                const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));  // hit lanes below this one
                const int dst = (hit ? below : n_hits + (lane - below)) << 2;  // a permutation: the hits first, in order
                const int line_c = __builtin_amdgcn_ds_permute(dst, line);
                const int lane_c = __builtin_amdgcn_ds_permute(dst, lane);
                const int fast_c = __builtin_amdgcn_ds_permute(dst, fast ? 1 : 0);
                const unsigned long long fast_o = __ballot(fast_c != 0);  // bit j: hit j is test-free
                if (lane < n_hits) stage[lane] = rec32_row[line_c];
                wave_sync();
                int k = 0;
                while (k < n_hits) {
                    const unsigned long long rest = ~(fast_o >> k);
                    const int run = rest ? __builtin_ctzll(rest) : 64;  // test-free hits from ordinal k on
                    for (int j = 0; j < run; j += 2) {
                        // two hits per trip; an odd last one runs a second time with a zero amplitude (adds exactly 0)
                        const bool two = j + 1 < run;
                        const WideRec32 ra = stage[k + j];
                        WideRec32 rb = stage[k + j + (two ? 1 : 0)];
                        rb.yk = two ? rb.yk : 0.f;
                        const float ca = region1_c0(base_h, ra), cb = region1_c0(base_h, rb);
#pragma unroll
                        for (int r = 0; r < R / 2; ++r) {
                            acc32[r] = region1_f32x2(acc32[r], dq[r], ca, ra);
                            acc32[r] = region1_f32x2(acc32[r], dq[r], cb, rb);
                        }
                    }
                    k += run;
                    if (k >= n_hits) break;
                    // hit k touches a window edge or the core (the fp64 branch below has the commentary)
                    const int b = __builtin_amdgcn_readlane(lane_c, k);
                    const int e = __builtin_amdgcn_readlane(line_c, k);
                    ++k;
                    const WideRec cur = rec_row[e];
                    const WideRec32 cur32 = rec32_row[e];
                    const int jlo = __builtin_amdgcn_readlane(sc.lo, b), jhi = __builtin_amdgcn_readlane(sc.hi, b);
                    const int jc = __builtin_amdgcn_readlane(sc.clo, b), jchi = __builtin_amdgcn_readlane(sc.chi, b);
                    const bool delegated = jc < 0;
                    const int jclo = delegated ? -jc - 1 : jc;
                    const RegionI k1 = {cur.yk, cur.cv, cur.cd};
                    float2v term32[R / 2];  // sum + fp32 rational of every point pair, once per hit
                    const float c32 = region1_c0(base_h, cur32);
#pragma unroll
                    for (int p = 0; p < R / 2; ++p) term32[p] = region1_f32x2(acc32[p], dq[p], c32, cur32);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int a = it0 + 64 * r, z = min(a + 64, it1);
                        if (z <= jlo || a >= jhi || a >= it1) continue;
                        const bool over_core = !(z <= jclo || a >= jchi);
                        if (!over_core || delegated) {
                            const int ir = idx0 + 64 * r;
                            const bool take = ir >= jlo && ir < jhi && !(over_core && ir >= jclo && ir < jchi);
                            acc32[r >> 1][r & 1] = take ? term32[r >> 1][r & 1] : acc32[r >> 1][r & 1];
                        } else {
                            const WideSlow sl = slow_row[e];
                            if (idx0 + 64 * r >= jlo && idx0 + 64 * r < jhi) {
                                const double nu_r = nus[(int64_t)idx0 + 64 * r];  // (a core kept by the wide role: rare; the exact frequency)
                                acc[r] = voigt_add(acc[r], nu_r - cur.lnu, cur.inv, sl.y, sl.amp, k1);
                            }
                        }
                    }
                }
                wave_sync();  // (the next chunk's records go into the same LDS)
                if (pending32 >= 48) {
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r] += (double)acc32[r >> 1][r & 1], acc32[r >> 1][r & 1] = 0.f;
                    pending32 = 0;
                }
                continue;
            }
            // walk the hits in ascending order.  The record is fetched with scalar loads straight into SGPRs, which the fp64
            // instructions take as operands (one each): no copy, no vector registers.  Its latency is not hidden by a
            // software prefetch — carrying a record across iterations makes the compiler park it in 12 VGPRs and move it
            // there with six v_mov per hit — but by the other waves of the SIMD.
            for (;;) {
                // A run of test-free hits (every point of the tile inside the window and in region I: the operations of
                // voigt_add's region-I branch, bit-identical, or the fp32 rational of the mixed-precision mode) is a loop of
                // its own — a single basic block, so the sums stay in their registers; as one arm of an if / else with the
                // general case the compiler forms them in fresh registers and copies them back, five v_mov_b64 per hit.
                while ((m & (0ull - m)) & mf) {
                    const int e = __builtin_amdgcn_readlane(line, __builtin_ctzll(m));
                    const unsigned long long m1 = m & (m - 1);
                    if constexpr (MIXED) {
                        // two test-free hits per trip: both 32-byte records are requested together (one wait instead of
                        // two — the fp32 arithmetic is short enough for the record fetch to show), the sums keep list order.
                        // When the next hit is not test-free the second evaluation runs on the first record again with a
                        // zero amplitude (a scalar select; adds exactly 0): the loop stays one basic block
                        const bool two = ((m1 & (0ull - m1)) & mf) != 0;
                        const int e1 = __builtin_amdgcn_readlane(line, __builtin_ctzll(two ? m1 : m));
                        const WideRec32 ra = rec32_row[e];
                        WideRec32 rb = rec32_row[e1];
                        rb.yk = two ? rb.yk : 0.f;
                        const float ca = region1_c0(base_h, ra), cb = region1_c0(base_h, rb);
#pragma unroll
                        for (int r = 0; r < R / 2; ++r) {
                            acc32[r] = region1_f32x2(acc32[r], dq[r], ca, ra);
                            acc32[r] = region1_f32x2(acc32[r], dq[r], cb, rb);
                        }
                        m = two ? (m1 & (m1 - 1)) : m1;
                        continue;
                    } else {
                        const WideRec cur = rec_row[e];
                        const RegionI k1 = {cur.yk, cur.cv, cur.cd};
                        const double c0 = (nu_base_vec - cur.lnu) * cur.inv;
                        region1_add_shared<R>(acc, dnu, cur.inv, c0, k1);
                    }
#ifdef SDX_WALK_STATS
                    ++st_fast;
#endif
                    m = m1;
                }
                if (!m) break;
                const int b = __builtin_ctzll(m);
                const int e = __builtin_amdgcn_readlane(line, b);
                const WideRec cur = rec_row[e];
                WideRec32 cur32;
                if (MIXED) cur32 = rec32_row[e];
                m &= m - 1;
#ifdef SDX_WALK_STATS
                ++st_general;
#endif
                {
                    // The tile touches a window edge or the core.  Per 64-point block r (scalar tests): outside the window:
                    // nothing; clear of the core: the region-I rational for the whole block, added where the point is
                    // inside the window; over a DELEGATED core: the same, minus the core points (the narrow role adds
                    // them); over a core kept here (wider than a narrow window): the full Voigt term per point.
                    const int jlo = __builtin_amdgcn_readlane(sc.lo, b), jhi = __builtin_amdgcn_readlane(sc.hi, b);
                    const int jc = __builtin_amdgcn_readlane(sc.clo, b), jchi = __builtin_amdgcn_readlane(sc.chi, b);
                    const bool delegated = jc < 0;
                    const int jclo = delegated ? -jc - 1 : jc;
                    const RegionI k1 = {cur.yk, cur.cv, cur.cd};
                    const double c0 = (nu_base_vec - cur.lnu) * cur.inv;
                    float2v term32[MIXED ? R / 2 : 1];  // mixed mode: sum + fp32 rational of every point pair, once per hit
                    if constexpr (MIXED) {
                        const float c32 = region1_c0(base_h, cur32);
#pragma unroll
                        for (int p = 0; p < R / 2; ++p) term32[p] = region1_f32x2(acc32[p], dq[p], c32, cur32);
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int a = it0 + 64 * r, z = min(a + 64, it1);
                        if (z <= jlo || a >= jhi || a >= it1) continue;
                        const bool over_core = !(z <= jclo || a >= jchi);
#ifdef SDX_WALK_STATS
                        if (!over_core || delegated) ++st_if; else ++st_slow;
#endif
                        if (!over_core || delegated) {
                            const int ir = idx0 + 64 * r;  // (points beyond the grid's end lie beyond every window: jhi <= N_nu)
                            const bool take = ir >= jlo && ir < jhi && !(over_core && ir >= jclo && ir < jchi);
                            if constexpr (MIXED) {  // the tolerance path evaluates window edges in fp32 too (pairs of blocks)
                                acc32[r >> 1][r & 1] = take ? term32[r >> 1][r & 1] : acc32[r >> 1][r & 1];
                            } else {
                                acc[r] = region1_add_if(acc[r], fma(dnu[MIXED ? 0 : r], cur.inv, c0), k1, take);
                            }
                        } else {
                            const WideSlow sl = slow_row[e];
                            if (idx0 + 64 * r >= jlo && idx0 + 64 * r < jhi) {
                                if constexpr (MIXED) {
                                    const double nu_r = nus[(int64_t)idx0 + 64 * r];  // (a core kept by the wide role: rare; the exact frequency)
                                    acc[r] = voigt_add(acc[r], nu_r - cur.lnu, cur.inv, sl.y, sl.amp, k1);
                                } else {
                                    // x from the tile offsets like every other point of the tile (a core kept here is wider than 128
                                    // points: |c0| stays below ~50, x is good to 1e-14 absolute) — forming nu_i = dnu + nu_base instead
                                    // made the compiler hoist four sums out of the walk and spill them to scratch in every wave
                                    acc[r] = voigt_add_x(acc[r], fma(dnu[MIXED ? 0 : r], cur.inv, c0), sl.y, sl.amp, k1);
                                }
                            }
                        }
                    }
                }
                if (!m) break;
            }
            if constexpr (MIXED) {
                // the fp32 running sums go into the fp64 sums once ~64 terms have gathered (checked per chunk, not per hit: a
                // branch around the fp64 accumulators inside the walk costs eight register copies per hit)
                if (pending32 >= 48) {
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r] += (double)acc32[r >> 1][r & 1], acc32[r >> 1][r & 1] = 0.f;
                    pending32 = 0;
                }
            }
        }
    }
#ifdef SDX_WALK_STATS
    const unsigned long long tq_last = wall_clock64();
#endif
    if constexpr (QUEUED) q_flush(q_count);
#ifdef SDX_WALK_STATS
    st_flush += (int)(wall_clock64() - tq_last);
#endif
    if constexpr (MIXED) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] += (double)acc32[r >> 1][r & 1];
    }
#ifdef SDX_WALK_STATS
    if (lane == 0) {  // (a device array read back by sdx_walk_stats_read: printf's host calls slowed the whole launch a thousandfold)
        const int slot = ((d * 512 + tile_idx) * 8 + split) & (kWalkStatSlots / 2 - 1);
        unsigned long long* const o = g_walk_stats + (size_t)slot * 8;
        o[0] = ((unsigned long long)d << 40) | ((unsigned long long)tile_idx << 8) | (unsigned long long)split | (1ull << 63);
        o[1] = st_t0, o[2] = wall_clock64(), o[3] = (unsigned long long)st_chunks | ((unsigned long long)(unsigned)st_flush << 32), o[4] = st_fast, o[5] = st_general, o[6] = st_if, o[7] = st_slow;
    }
#endif
    if constexpr (DEFER) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc_out[r] = acc[r];
    } else {
        wide_reduce_and_store<R, kLdsStride>(split, n_split, acc, idx0, s0, s1, lds_all, plane, pld, d);
    }
}

// Long line lists: two stable compactions of the per-line classes in two small launches (per-block counts, then every
// block sums the counts before it — a few hundred integers — and scatters its own lines):
//   hlist  the lines whose widest window exceeds kMediumHalfWidth (they may reach any tile: every tile visits them all)
//   wlist  the other lines with a wide window at some depth, + wrank[l] = wlist entries below line l
constexpr int kHlistBlock = 1024;
__device__ __forceinline__ int line_class(int whw) { return whw > kMediumHalfWidth ? 2 : (whw > kNarrowHalfWidth ? 1 : 0); }

// (xlist: the class-1 lines of [sel[0], sel[1]) outside the blocks of pre_lines consecutive lines that cover [sel[2], sel[3]))
__device__ __forceinline__ bool in_xlist(int64_t l, int cls, const int* __restrict__ sel, int pre_lines)
{
    if (!sel || cls != 1) return false;
    const int ra = sel[2] / pre_lines * pre_lines, rb = (sel[3] + pre_lines - 1) / pre_lines * pre_lines;
    return l >= sel[0] && l < sel[1] && !(sel[3] > sel[2] && l >= ra && l < rb);
}

// the window half-width of the largest (gamma + dw) alpha of a line (:561-575; NaN or negative: the 10-point floor)
__device__ __forceinline__ int half_width_of(double m, double d_nu, int64_t n_nu)
{
    const double pixels = mul_rn(m / d_nu, 20.0);
    const double forced = pixels > 10.0 ? pixels : 10.0;
    return (int)(forced >= (double)n_nu ? n_nu : (int64_t)forced);
}
// (culled runs: the class comes from the classification pass's largest (gamma + dw) alpha per line and the grid spacing)
struct ClassSource {
    const int* whw_max;       // per line, from a full pre-pass; or nullptr:
    const double* m_max;      // per line, from k_classify
    const double* dnu_partial;
    int n_partial;
    int64_t n_nu;
};
__device__ __forceinline__ int class_of_line(const ClassSource& cs, int64_t l, double d_nu)
{
    return line_class(cs.whw_max ? cs.whw_max[l] : half_width_of(cs.m_max[l], d_nu, cs.n_nu));
}
__global__ __launch_bounds__(kHlistBlock) void k_hlist_count(int64_t n_lines, ClassSource cs, int* __restrict__ block_cnt,
                                                            const int* __restrict__ sel, int pre_lines, int* __restrict__ ticket)
{
    __shared__ int s_wave[3][kHlistBlock / 64];
    __shared__ double s_dnu[kHlistBlock / 64];
    if (ticket && blockIdx.x == 0 && threadIdx.x < 8) ticket[threadIdx.x] = 0;  // (the work counters of the pre-pass launch that follows)
    const double d_nu = cs.whw_max ? 0.0 : block_dnu(cs.dnu_partial, cs.n_partial, s_dnu);
    const int64_t l = (int64_t)blockIdx.x * kHlistBlock + threadIdx.x;
    const int cls = l < n_lines ? class_of_line(cs, l, d_nu) : 0;
    const unsigned long long mh = __ballot(cls == 2), mw = __ballot(cls == 1), mx = __ballot(in_xlist(l, cls, sel, pre_lines));
    if ((threadIdx.x & 63) == 0)
        s_wave[0][threadIdx.x >> 6] = __popcll(mh), s_wave[1][threadIdx.x >> 6] = __popcll(mw), s_wave[2][threadIdx.x >> 6] = __popcll(mx);
    __syncthreads();
    if (threadIdx.x < 3) {
        int tot = 0;
        for (int k = 0; k < kHlistBlock / 64; ++k) tot += s_wave[threadIdx.x][k];
        block_cnt[3 * blockIdx.x + threadIdx.x] = tot;
    }
}
__global__ __launch_bounds__(kHlistBlock) void k_hlist_scatter(int64_t n_lines, ClassSource cs, const int* __restrict__ block_cnt,
                                                              int* __restrict__ hlist, int* __restrict__ wlist, int* __restrict__ wrank,
                                                              int* __restrict__ hcount, int* __restrict__ xlist, const int* __restrict__ sel,
                                                              int pre_lines)
{
    __shared__ int s_wave[3][kHlistBlock / 64];
    __shared__ int s_red[3][kHlistBlock / 64];
    __shared__ double s_dnu[kHlistBlock / 64];
    const double d_nu = cs.whw_max ? 0.0 : block_dnu(cs.dnu_partial, cs.n_partial, s_dnu);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int bh = 0, bw = 0, bx = 0;
    for (int k = threadIdx.x; k < (int)blockIdx.x; k += kHlistBlock) bh += block_cnt[3 * k], bw += block_cnt[3 * k + 1], bx += block_cnt[3 * k + 2];
    for (int off = 32; off > 0; off >>= 1) bh += __shfl_xor(bh, off), bw += __shfl_xor(bw, off), bx += __shfl_xor(bx, off);
    const int64_t l = (int64_t)blockIdx.x * kHlistBlock + threadIdx.x;
    const int cls = l < n_lines ? class_of_line(cs, l, d_nu) : 0;
    const bool isx = in_xlist(l, cls, sel, pre_lines);
    const unsigned long long mh = __ballot(cls == 2), mw = __ballot(cls == 1), mx = __ballot(isx);
    if (lane == 0) {
        s_red[0][wave] = bh, s_red[1][wave] = bw, s_red[2][wave] = bx;
        s_wave[0][wave] = __popcll(mh), s_wave[1][wave] = __popcll(mw), s_wave[2][wave] = __popcll(mx);
    }
    __syncthreads();
    int base_h = 0, base_w = 0, base_x = 0;
    for (int k = 0; k < kHlistBlock / 64; ++k) base_h += s_red[0][k], base_w += s_red[1][k], base_x += s_red[2][k];
    for (int k = 0; k < wave; ++k) base_h += s_wave[0][k], base_w += s_wave[1][k], base_x += s_wave[2][k];
    const unsigned long long below = (1ull << lane) - 1ull;
    const int pos_h = base_h + __popcll(mh & below), pos_w = base_w + __popcll(mw & below), pos_x = base_x + __popcll(mx & below);
    if (cls == 2) hlist[pos_h] = (int)l;
    if (cls == 1) wlist[pos_w] = (int)l;
    if (isx) xlist[pos_x] = (int)l;
    if (l < n_lines) wrank[l] = pos_w;
    if (l == n_lines - 1) {
        wrank[n_lines] = pos_w + (cls == 1);
        hcount[0] = pos_h + (cls == 2);
        hcount[1] = pos_w + (cls == 1);
        hcount[2] = pos_x + (isx ? 1 : 0);
    }
}

// hscan[d][k] = wscan[d][hlist[k]]: every tile scans ALL the huge lines; gathering their 16-byte scan words through the
// list costs a 64-byte sector each.  Copied once into list order they are read as contiguous kilobytes.
__global__ __launch_bounds__(kBlock) void k_hscan(int n_depth, int64_t n_lines, const int* __restrict__ hlist, const int* __restrict__ hcount,
                                                  const WideScan* __restrict__ wscan, WideScan* __restrict__ hscan)
{
    const int n_h = hcount[0];
    const int d = blockIdx.y;
    for (int k = blockIdx.x * kBlock + threadIdx.x; k < n_h; k += gridDim.x * kBlock)
        hscan[(size_t)d * n_lines + k] = wscan[(size_t)d * n_lines + hlist[k]];
}

// sel[0..1] = [la, lb): the lines whose centre c satisfies begin - H < c < end + H for the columns [begin, end) of the shard's
// tiles and H = kMediumHalfWidth; sel[2..3] = [na, nb): the same for the shard's own columns with H = 2 kNarrowReach
// (centre_l = #{i : nus[i] >= line_nu_l}; lines ascend in frequency, so centres descend with the line index)
__device__ __forceinline__ void shard_range(int64_t n_nu, const double* __restrict__ nus, int64_t n_lines, const double* __restrict__ line_nus,
                                            int64_t nu_begin, int64_t nu_count, int* __restrict__ sel)
{
    // one WAVE per bound (the first four waves of the block), each a 64-ary search: three round trips to memory for 1.5e5 lines
    // where a thread's bisection makes seventeen — in the two-collective mode the classification launch is a few microseconds of
    // streaming, and the four chains of dependent loads were what it waited for (24 - 29 us on an eighth of S-c3)
    const int which = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (which >= 4) return;
    // lines with centre >= p  <=>  line_nu <= nus[p - 1]: their number is cnt_ge[p]
    const int64_t H = which < 2 ? kMediumHalfWidth : 2 * kNarrowReach;
    // ... [la, lb) for the shard's columns rounded out to whole TILES of the wide role (kMaxTile covers every tile width): a tile
    // cut by the shard boundary is walked whole, and its candidate list — with it the points at which the fp32-mixed mode folds its
    // running sums into the fp64 ones, which count the hits of the whole tile — must be the unsharded run's for the shard to
    // reproduce that run's bits in BOTH precisions (tests/test_gpu_long_random.py found mixed-mode shards a few 1e-8 apart)
    const int64_t ext = which < 2 ? kMaxTile - 1 : 0;
    const int64_t pa = max(nu_begin - ext - H + 1, (int64_t)0), pb = min(nu_begin + nu_count + ext + H - 1, n_nu);
    const int64_t p = (which & 1) == 0 ? pb + 1 : pa;  // sel[0] = cnt_ge[pb + 1], sel[1] = cnt_ge[pa]
    int64_t cnt;
    if (p == 0) cnt = n_lines;
    else if (p >= n_nu + 1) cnt = 0;
    else {
        const double v = nus[p - 1];
        int64_t lo = 0, hi = n_lines;  // the first l with line_nus[l] > v lies in [lo, hi]
        while (hi > lo) {
            const int64_t step = (hi - lo + 63) / 64;
            const int64_t idx = lo + (int64_t)lane * step;  // the lanes' probes ascend: `le` is true on a prefix of them
            const bool le = idx < hi && line_nus[idx] <= v;
            const int k = __popcll(__ballot(le));
            const int64_t base = lo;
            lo = k > 0 ? base + (int64_t)(k - 1) * step + 1 : base;
            hi = min(base + (int64_t)k * step, hi);
        }
        cnt = lo;
    }
    if (lane == 0) sel[which] = (int)cnt;
}

// Frequency-sharded runs of long lists, stage A: how wide the widest window of every line is (over all depths) — which lines can
// reach any column (-> hlist) or, from just outside the shard's own line range, its edge columns (-> xlist) — before the full
// pre-pass runs on the lines the shard needs.  The half-width of the window rule (:561-575), int(max(10, (gamma + dw) alpha / d_nu
// * 20)), is a non-decreasing function of m = (gamma + dw) alpha, so the widest window of a line is the window of its LARGEST m:
// this pass streams the dense inputs once and leaves max_d m per line; the list kernels apply the rule (they, not this pass,
// need the grid spacing — whose partial maxima the FIRST blocks of this launch compute on the side: one launch fewer per step).
// One wave per line at a time, lane <-> depth: a line's 56 values are one contiguous 448-byte request per array, the maximum is
// a wave reduction, the result a plain store — no atomics, nothing to clear beforehand.  (Round 3 accumulated integer half-widths
// with atomicMax behind a pre-filter that needed d_nu: k_dnu_partial had to run, and clear the maxima, before every step.)
// NaN terms are ignored (their window is the 10-point floor: never the widest).
constexpr int kClsLinesPerWave = 4;  // lines in flight per wave: 12 independent loads per lane
__device__ __forceinline__ void classify_block(const int bid, const int n_blocks, int n_depth, const int64_t cls_begin, const int64_t cls_end,
                                               const double* __restrict__ doppler, const double* __restrict__ gammas, int gamma_cols,
                                               const double* __restrict__ alphas, double* __restrict__ m_max)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    // the lines [cls_begin, cls_end) — the whole list, or this rank's share of it (two-collective mode: the ranks exchange their
    // shares of m_max) — in contiguous runs per block; a block's waves take consecutive groups of kClsLinesPerWave lines
    const int64_t per_block = (cls_end - cls_begin + n_blocks - 1) / n_blocks;
    const int64_t l0 = cls_begin + (int64_t)bid * per_block, l1 = min(l0 + per_block, cls_end);
    for (int64_t base = l0 + (int64_t)wave * kClsLinesPerWave; base < l1; base += (int64_t)n_waves * kClsLinesPerWave) {
        double m[kClsLinesPerWave];
#pragma unroll
        for (int j = 0; j < kClsLinesPerWave; ++j) m[j] = -INFINITY;
        for (int d = lane; d < n_depth; d += 64) {
            double dw[kClsLinesPerWave], al[kClsLinesPerWave], g[kClsLinesPerWave];
#pragma unroll
            for (int j = 0; j < kClsLinesPerWave; ++j) {
                const int64_t l = min(base + j, l1 - 1);  // (clamped: a repeated line changes nothing)
                dw[j] = doppler[l * n_depth + d];
                al[j] = alphas[l * n_depth + d];
                g[j] = gamma_cols > 1 ? gammas[l * gamma_cols + d] : gammas[l];
            }
#pragma unroll
            for (int j = 0; j < kClsLinesPerWave; ++j) m[j] = fmax(m[j], mul_rn(add_rn(g[j], dw[j]), al[j]));  // (fmax drops a NaN)
        }
#pragma unroll
        for (int j = 0; j < kClsLinesPerWave; ++j) {
            for (int off = 32; off > 0; off >>= 1) m[j] = fmax(m[j], __shfl_xor(m[j], off));
            if (lane == 0 && base + j < l1) m_max[base + j] = m[j];
        }
    }
}
// this block's share of max(diff(nus)) -> partial[bid]  (blocks [0, n_dnu) of the classification launches)
__device__ __forceinline__ void dnu_partial_block(const int bid, const int n_blocks, int64_t n_nu, const double* __restrict__ nus,
                                                  double* __restrict__ partial, double* s_red, const FarReq& far)
{
    far_ranges_block(bid, n_blocks, n_nu, nus, far);
    grid_sample_block(bid, n_blocks, n_nu, nus, partial + kGridSampleOffset);
    // four independent pairs of loads in flight per thread: the scan is a chain of round trips, not a stream (the grid sits in L2)
    double m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
    const int64_t step = (int64_t)n_blocks * blockDim.x;
    int64_t i = (int64_t)bid * blockDim.x + threadIdx.x;
    for (; i + 3 * step + 1 < n_nu; i += 4 * step) {
        const double a0 = nus[i], b0 = nus[i + 1], a1 = nus[i + step], b1 = nus[i + step + 1];
        const double a2 = nus[i + 2 * step], b2 = nus[i + 2 * step + 1], a3 = nus[i + 3 * step], b3 = nus[i + 3 * step + 1];
        m0 = fmax(m0, b0 - a0), m1 = fmax(m1, b1 - a1), m2 = fmax(m2, b2 - a2), m3 = fmax(m3, b3 - a3);
    }
    for (; i + 1 < n_nu; i += step) m0 = fmax(m0, nus[i + 1] - nus[i]);
    double m = fmax(fmax(m0, m1), fmax(m2, m3));
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmax(m, s_red[w]);
        partial[bid] = m;
    }
}

__global__ __launch_bounds__(kBlock) void k_classify(int n_dnu, int n_depth, int64_t n_nu, int64_t n_lines, double* __restrict__ dnu_partial,
                                                     const double* __restrict__ doppler, const double* __restrict__ gammas,
                                                     int gamma_cols, const double* __restrict__ alphas, double* __restrict__ m_max,
                                                     const double* __restrict__ nus, const double* __restrict__ line_nus, int64_t nu_begin,
                                                     int64_t nu_count, int* __restrict__ sel, int64_t cls_begin, int64_t cls_end, FarReq far)
{
    __shared__ double s_red[kBlock / 64];
    const int b = blockIdx.x;
    if (b < n_dnu) {
        dnu_partial_block(b, n_dnu, n_nu, nus, dnu_partial, s_red, far);
        return;
    }
    // four threads of the LAST block find the shard's line ranges on the side (four binary searches: chains of dependent loads
    // that vanish behind this stream)
    if (sel && b == (int)gridDim.x - 1) shard_range(n_nu, nus, n_lines, line_nus, nu_begin, nu_count, sel);
    classify_block(b - n_dnu, (int)gridDim.x - n_dnu, n_depth, cls_begin, cls_end, doppler, gammas, gamma_cols, alphas, m_max);
}


// Narrow windows (half-width <= kNarrowHalfWidth, e.g. the reference's 10-pixel floor for weak lines, :565-567):
// a 256-point tile would be almost empty for them.  Here a wave owns ONE frequency and its lanes are the depth
// points (lane <-> depth, line-major parameter arrays so the loads coalesce): a weak line covers either all
// depths of that frequency or none, and all lanes share x = (nu_i - nu_l)/doppler up to the slowly varying Doppler
// width, so the Faddeeva region rarely diverges inside a wave.  The candidate lines of a frequency are the
// contiguous index range whose centre lies within kNarrowHalfWidth of it, read from cnt_ge; each lane walks that
// range in ascending line order, tests its own window and accumulates in a register.  Deterministic, no atomics.
__device__ __forceinline__ void line_narrow_wave(const int64_t i, const int depth_chunk, int n_depth, int64_t n_nu, const double* __restrict__ nus,
                                                        int64_t nu_begin, int64_t nu_count, int64_t n_lines,
                                                        const double* __restrict__ line_nus, LineWork w,
                                                        double* __restrict__ plane, int64_t pld)
{
    const int lane = threadIdx.x & 63;
    if (i >= nu_begin + nu_count) return;
    const int d = depth_chunk * 64 + lane;
    const bool valid = d < n_depth;
    const int dc = valid ? d : n_depth - 1;
    const unsigned dcu = (unsigned)dc;
    const bool one_gamma = w.narrow_raw == 1;  // (raw inputs with gammas (N_l, 1): the line's one value for every depth)
    const int ii = (int)i;
    // lines with centre c in [i - H + 1, i + H]
    const int64_t pa = max(i - kNarrowReach + 1, (int64_t)0);
    const int64_t pb = min(i + kNarrowReach, n_nu);
    const int la = __builtin_amdgcn_readfirstlane(w.cnt_ge[pb + 1]);
    const int lb = __builtin_amdgcn_readfirstlane(w.cnt_ge[pa]);
    const double nu_i = nus[i];
    double acc = 0.0;
#ifdef SDX_WALK_STATS
    const unsigned long long st_t0 = wall_clock64();
    int st_chunks = 0, st_rel = 0, st_eval = 0;
#endif
    for (int base = la; base < lb; base += 64) {
        // lanes test 64 candidate lines at once against this frequency (per-line bound), then the wave visits
        // only the relevant ones, in ascending line order
        const int lc = base + lane;
        bool rel = false;
        int c = 0;
        if (lc < lb) {
            const int hwm = w.nhw_max[lc];
            c = w.centre[lc];
            rel = hwm > 0 && ii >= c - hwm && ii < c + hwm;
        }
        unsigned long long m = __ballot(rel);
#ifdef SDX_WALK_STATS
        ++st_chunks, st_rel += __popcll(m);
#endif
        // the parameters of the NEXT relevant line are requested before the current one is evaluated (all six loads at
        // once, used or not): one global-memory round trip per line hides behind the previous line's arithmetic.  (Two lines in
        // flight — pairs — and the next chunk's candidates requested ahead were measured in round 4: 24.4 against 24.2 us for the
        // role alone at S-c2, 37.1 against 37.0 together: a wave's line takes ~1 us because seven waves share the SIMD's
        // arithmetic, not because it waits for memory.)
        int h = 0, cl = 0;  // this depth's half-width, the line's centre (scalar)
        double y = 0.0, amp = 0.0, inv = 0.0, lnu = 0.0;
        // addresses: a per-line base (uniform: scalar arithmetic) + the lane's depth as an unsigned 32-bit offset
        if (m) {
            const int bit = __builtin_ctzll(m);
            const int l = base + bit;
            const size_t ob = (size_t)l * n_depth;
            cl = __builtin_amdgcn_readlane(c, bit);
            h = (w.nhw + ob)[dcu], y = (w.n_y + (one_gamma ? (size_t)l : ob))[one_gamma ? 0u : dcu], amp = (w.n_amp + ob)[dcu], inv = (w.n_inv + ob)[dcu], lnu = line_nus[l];
        }
        while (m) {
            m &= m - 1;
            int h_n = 0, cl_n = 0;
            double y_n = 0.0, amp_n = 0.0, inv_n = 0.0, lnu_n = 0.0;
            if (m) {
                const int bit = __builtin_ctzll(m);
                const int l = base + bit;
                const size_t ob = (size_t)l * n_depth;
                cl_n = __builtin_amdgcn_readlane(c, bit);
                h_n = (w.nhw + ob)[dcu], y_n = (w.n_y + (one_gamma ? (size_t)l : ob))[one_gamma ? 0u : dcu], amp_n = (w.n_amp + ob)[dcu], inv_n = (w.n_inv + ob)[dcu], lnu_n = line_nus[l];
            }
#ifdef SDX_WALK_STATS
            if (__ballot(valid && ii >= cl - h && ii < cl + h)) ++st_eval;
#endif
            if (valid && ii >= cl - h && ii < cl + h) {  // (h = 0: empty)
                if (w.narrow_raw) narrow_params(inv, y, amp, inv, y, amp);  // (inv, y, amp hold dw, gamma, alpha)
                const RegionI k1 = region1_setup(y, amp);
                acc = voigt_add(acc, nu_i - lnu, inv, y, amp, k1);
            }
            h = h_n, cl = cl_n, y = y_n, amp = amp_n, inv = inv_n, lnu = lnu_n;
        }
    }
#ifdef SDX_WALK_STATS
    if (lane == 0) {
        const int slot = (int)(kWalkStatSlots / 2 + (i & (kWalkStatSlots / 2 - 1)));
        unsigned long long* const o = g_walk_stats + (size_t)slot * 8;
        o[0] = (unsigned long long)i | (3ull << 62);
        o[1] = st_t0, o[2] = wall_clock64(), o[3] = st_chunks, o[4] = st_rel, o[5] = st_eval, o[6] = 0, o[7] = 0;
    }
#endif
    if (valid) plane[(size_t)d * pld + (i - nu_begin)] = acc;
}

// The same walk for F CONSECUTIVE frequencies per wave (F = 2, 4; groups aligned to the global grid index): the records of a
// visited line — five coalesced loads, 1.9 KB per wave — serve up to F evaluations instead of one, the candidate test and the
// region-I constants are shared, and a lane has F independent evaluations in flight.  At 1e6 lines the one-frequency walk
// fetched 16 GB per launch through its record loads (each of the ~20 frequencies of a weak line's window loaded them again,
// and 900 waves per XCD, each with its own 170 lines, do not fit the L2); a group fetches a line's records once.  Every
// frequency still adds its lines in ascending line order: the bits are those of the one-frequency walk, whatever F — which
// may therefore depend on the size of the launch (small grids keep F = 1: they need the waves).
template <int F>
__device__ __forceinline__ void line_narrow_group(const int64_t i0, const int depth_chunk, int n_depth, int64_t n_nu, const double* __restrict__ nus,
                                                  int64_t nu_begin, int64_t nu_count, int64_t n_lines, const double* __restrict__ line_nus,
                                                  LineWork w, double* __restrict__ plane, int64_t pld)
{
    const int lane = threadIdx.x & 63;
    const int d = depth_chunk * 64 + lane;
    const bool valid = d < n_depth;
    const int dc = valid ? d : n_depth - 1;
    const unsigned dcu = (unsigned)dc;
    const bool one_gamma = w.narrow_raw == 1;  // (raw inputs with gammas (N_l, 1): the line's one value for every depth)
    const int ia = (int)i0;
    // lines with centre c in [i0 - H + 1, i0 + F - 1 + H]
    const int64_t pa = max(i0 - kNarrowReach + 1, (int64_t)0);
    const int64_t pb = min(i0 + F - 1 + kNarrowReach, n_nu);
    const int la = __builtin_amdgcn_readfirstlane(w.cnt_ge[pb + 1]);
    const int lb = __builtin_amdgcn_readfirstlane(w.cnt_ge[pa]);
    double nu_k[F], acc[F];
    bool act[F];  // the frequency belongs to this launch's columns (wave-uniform)
#pragma unroll
    for (int k = 0; k < F; ++k) {
        act[k] = i0 + k >= nu_begin && i0 + k < nu_begin + nu_count;
        nu_k[k] = nus[min(i0 + k, n_nu - 1)];
        acc[k] = 0.0;
    }
    for (int base = la; base < lb; base += 64) {
        const int lc = base + lane;
        bool rel = false;
        int c = 0;
        if (lc < lb) {
            const int hwm = w.nhw_max[lc];
            c = w.centre[lc];
            rel = hwm > 0 && ia + (F - 1) >= c - hwm && ia < c + hwm;
        }
        unsigned long long m = __ballot(rel);
        int h = 0, cl = 0;
        double y = 0.0, amp = 0.0, inv = 0.0, lnu = 0.0;
        if (m) {
            const int bit = __builtin_ctzll(m);
            const int l = base + bit;
            const size_t ob = (size_t)l * n_depth;
            cl = __builtin_amdgcn_readlane(c, bit);
            h = (w.nhw + ob)[dcu], y = (w.n_y + (one_gamma ? (size_t)l : ob))[one_gamma ? 0u : dcu], amp = (w.n_amp + ob)[dcu], inv = (w.n_inv + ob)[dcu], lnu = line_nus[l];
        }
        while (m) {
            m &= m - 1;
            int h_n = 0, cl_n = 0;
            double y_n = 0.0, amp_n = 0.0, inv_n = 0.0, lnu_n = 0.0;
            if (m) {
                const int bit = __builtin_ctzll(m);
                const int l = base + bit;
                const size_t ob = (size_t)l * n_depth;
                cl_n = __builtin_amdgcn_readlane(c, bit);
                h_n = (w.nhw + ob)[dcu], y_n = (w.n_y + (one_gamma ? (size_t)l : ob))[one_gamma ? 0u : dcu], amp_n = (w.n_amp + ob)[dcu], inv_n = (w.n_inv + ob)[dcu], lnu_n = line_nus[l];
            }
            const int lo = cl - h, hi = cl + h;  // (h = 0: empty; the clamp to the grid is implied by ia + k being a grid index)
            if (valid && ia + (F - 1) >= lo && ia < hi) {
                if (w.narrow_raw) narrow_params(inv, y, amp, inv, y, amp);  // (inv, y, amp hold dw, gamma, alpha)
                const RegionI k1 = region1_setup(y, amp);
#pragma unroll
                for (int k = 0; k < F; ++k)
                    if (act[k] && ia + k >= lo && ia + k < hi) acc[k] = voigt_add(acc[k], nu_k[k] - lnu, inv, y, amp, k1);
            }
            h = h_n, cl = cl_n, y = y_n, amp = amp_n, inv = inv_n, lnu = lnu_n;
        }
    }
#pragma unroll
    for (int k = 0; k < F; ++k)
        if (valid && act[k]) plane[(size_t)d * pld + (i0 + k - nu_begin)] = acc[k];
}

// The narrow role of DENSE long lists (at least one line per two grid points, four line subsets): the four waves of a workgroup
// share ONE group of F consecutive frequencies and split its candidate lines — wave s takes the lines l with (l / 16) % 4 == s,
// sixteen-line runs of the LIST going round the waves, so every wave gets a quarter of every stretch of the list — and the four
// partial sums meet in LDS, where wave 0 adds them in subset order.  A wave still fetches a visited line's records once for up to
// F evaluations (what F = 4 frequencies per wave saved: a third of the role's instructions and three quarters of its fetch), but
// there are as many waves as with one frequency per wave: a frequency SHARD of 11 000 - 21 000 columns fills the chip (round 4 had
// to fall back to one frequency per wave there: 3 000 four-times-longer waves are a launch's tail).  Which lines share a partial
// sum is a property of the list (the run index of a line), so shards reproduce the unsharded bits; the sums differ from the
// sequential walk's in the last bits (four partial sums instead of one).
template <int F>
__device__ __forceinline__ void line_narrow_subsets(const int64_t i0, const int depth_chunk, const int subset, int n_depth, int64_t n_nu,
                                                    const double* __restrict__ nus, int64_t nu_begin, int64_t nu_count,
                                                    const double* __restrict__ line_nus, LineWork w, double* __restrict__ acc_out)
{
    const int lane = threadIdx.x & 63;
    const int d = depth_chunk * 64 + lane;
    const bool valid = d < n_depth;
    const int dc = valid ? d : n_depth - 1;
    const unsigned dcu = (unsigned)dc;
    const bool one_gamma = w.narrow_raw == 1;  // (raw inputs with gammas (N_l, 1): the line's one value for every depth)
    const int ia = (int)i0;
    // lines with centre c in [i0 - H + 1, i0 + F - 1 + H]
    const int64_t pa = max(i0 - kNarrowReach + 1, (int64_t)0);
    const int64_t pb = min(i0 + F - 1 + kNarrowReach, n_nu);
    const int la = __builtin_amdgcn_readfirstlane(w.cnt_ge[pb + 1]);
    const int lb = __builtin_amdgcn_readfirstlane(w.cnt_ge[pa]);
    double nu_k[F], acc[F];
    bool act[F];  // the frequency belongs to this launch's columns (wave-uniform)
#pragma unroll
    for (int k = 0; k < F; ++k) {
        act[k] = i0 + k >= nu_begin && i0 + k < nu_begin + nu_count;
        nu_k[k] = nus[min(i0 + k, n_nu - 1)];
        acc[k] = 0.0;
    }
    // lane -> candidate: 64 lines of this subset per trip, out of a stretch of 256 aligned to the LIST (lanes ascend with the line index)
    const int lane_off = ((lane >> 4) << 6) + (subset << 4) + (lane & 15);
    for (int base = la & ~255; base < lb; base += 256) {
        const int lc = base + lane_off;
        bool rel = false;
        int c = 0;
        if (lc >= la && lc < lb) {
            const int hwm = w.nhw_max[lc];
            c = w.centre[lc];
            rel = hwm > 0 && ia + (F - 1) >= c - hwm && ia < c + hwm;
        }
        unsigned long long m = __ballot(rel);
        // the parameters of the NEXT relevant line are requested before the current one is evaluated
        int h = 0, cl = 0;
        double y = 0.0, amp = 0.0, inv = 0.0, lnu = 0.0;
        if (m) {
            const int bit = __builtin_ctzll(m);
            const int l = base + ((bit >> 4) << 6) + (subset << 4) + (bit & 15);
            const size_t ob = (size_t)l * n_depth;
            cl = __builtin_amdgcn_readlane(c, bit);
            h = (w.nhw + ob)[dcu], y = (w.n_y + (one_gamma ? (size_t)l : ob))[one_gamma ? 0u : dcu], amp = (w.n_amp + ob)[dcu], inv = (w.n_inv + ob)[dcu], lnu = line_nus[l];
        }
        while (m) {
            m &= m - 1;
            int h_n = 0, cl_n = 0;
            double y_n = 0.0, amp_n = 0.0, inv_n = 0.0, lnu_n = 0.0;
            if (m) {
                const int bit = __builtin_ctzll(m);
                const int l = base + ((bit >> 4) << 6) + (subset << 4) + (bit & 15);
                const size_t ob = (size_t)l * n_depth;
                cl_n = __builtin_amdgcn_readlane(c, bit);
                h_n = (w.nhw + ob)[dcu], y_n = (w.n_y + (one_gamma ? (size_t)l : ob))[one_gamma ? 0u : dcu], amp_n = (w.n_amp + ob)[dcu], inv_n = (w.n_inv + ob)[dcu], lnu_n = line_nus[l];
            }
            const int lo = cl - h, hi = cl + h;  // (h = 0: empty; the clamp to the grid is implied by ia + k being a grid index)
            if (valid && ia + (F - 1) >= lo && ia < hi) {
                if (w.narrow_raw) narrow_params(inv, y, amp, inv, y, amp);  // (inv, y, amp hold dw, gamma, alpha)
                const RegionI k1 = region1_setup(y, amp);
#pragma unroll
                for (int k = 0; k < F; ++k)
                    if (act[k] && ia + k >= lo && ia + k < hi) acc[k] = voigt_add(acc[k], nu_k[k] - lnu, inv, y, amp, k1);
            }
            h = h_n, cl = cl_n, y = y_n, amp = amp_n, inv = inv_n, lnu = lnu_n;
        }
    }
#pragma unroll
    for (int k = 0; k < F; ++k) acc_out[k] = acc[k];  // (the four subsets of a group meet in the kernel's common reduction)
}

// The narrow role of the mixed-precision mode: the same walk, every term evaluated by voigt_add32 (packed fp32, all four
// regions) from fp32 records; the frequency difference comes from the hi + lo float pairs of both frequencies (exact to
// 2^-48 of the frequency, i.e. ~1e-7 of a narrow window's reach), the terms of one 64-candidate chunk gather in an fp32
// sum that goes into the fp64 sum once per chunk.
__device__ __forceinline__ void line_narrow_wave32(const int64_t i, const int depth_chunk, int n_depth, int64_t n_nu, const double* __restrict__ nus,
                                                          int64_t nu_begin, int64_t nu_count, LineWork w, double* __restrict__ plane, int64_t pld)
{
    const int lane = threadIdx.x & 63;
    if (i >= nu_begin + nu_count) return;
    const int d = depth_chunk * 64 + lane;
    const bool valid = d < n_depth;
    const int dc = valid ? d : n_depth - 1;
    const int ii = (int)i;
    const int64_t pa = max(i - kNarrowReach + 1, (int64_t)0);
    const int64_t pb = min(i + kNarrowReach, n_nu);
    const int la = __builtin_amdgcn_readfirstlane(w.cnt_ge[pb + 1]);
    const int lb = __builtin_amdgcn_readfirstlane(w.cnt_ge[pa]);
    const double nu_i = nus[i];
    const float nih = (float)nu_i, nil = (float)(nu_i - (double)nih);
    double acc = 0.0;
    // (chunks of 64 candidates aligned to the LIST — base = 64 k — not to this frequency's first candidate: the fp32 sum of a chunk
    // is folded into the fp64 sum once per chunk, and which terms share a chunk must not depend on where the candidate range of a
    // frequency, or of a group of F frequencies, happens to start — F follows the launch's width, a shard's differs from the whole grid's)
    for (int base = la & ~63; base < lb; base += 64) {
        const int lc = base + lane;
        bool rel = false;
        int c = 0;
        if (lc >= la && lc < lb) {
            const int hwm = w.nhw_max[lc];
            c = w.centre[lc];
            rel = hwm > 0 && ii >= c - hwm && ii < c + hwm;
        }
        unsigned long long m = __ballot(rel);
        float acc32 = 0.f;
        int h = 0, cl = 0;
        float y = 0.f, amp = 0.f, inv = 0.f;
        float2v lnu = {0.f, 0.f};
        if (m) {
            const int bit = __builtin_ctzll(m);
            const int l = base + bit;
            const size_t o = (size_t)l * n_depth + dc;
            cl = __builtin_amdgcn_readlane(c, bit);
            h = w.nhw[o], y = w.n_y32[o], amp = w.n_amp32[o], inv = w.n_inv32[o], lnu = w.lnu32[l];
        }
        while (m) {
            m &= m - 1;
            int h_n = 0, cl_n = 0;
            float y_n = 0.f, amp_n = 0.f, inv_n = 0.f;
            float2v lnu_n = {0.f, 0.f};
            if (m) {
                const int bit = __builtin_ctzll(m);
                const int l = base + bit;
                const size_t o = (size_t)l * n_depth + dc;
                cl_n = __builtin_amdgcn_readlane(c, bit);
                h_n = w.nhw[o], y_n = w.n_y32[o], amp_n = w.n_amp32[o], inv_n = w.n_inv32[o], lnu_n = w.lnu32[l];
            }
            if (valid && ii >= cl - h && ii < cl + h) acc32 = voigt_add32(acc32, ((nih - lnu.x) + (nil - lnu.y)) * inv, y, amp);
            h = h_n, cl = cl_n, y = y_n, amp = amp_n, inv = inv_n, lnu = lnu_n;
        }
        acc += (double)acc32;
    }
    if (valid) plane[(size_t)d * pld + (i - nu_begin)] = acc;
}

// ... and for F consecutive frequencies per wave, like line_narrow_group: a visited line's fp32 records serve up to F evaluations.
template <int F>
__device__ __forceinline__ void line_narrow_group32(const int64_t i0, const int depth_chunk, int n_depth, int64_t n_nu, const double* __restrict__ nus,
                                                    int64_t nu_begin, int64_t nu_count, LineWork w, double* __restrict__ plane, int64_t pld)
{
    const int lane = threadIdx.x & 63;
    const int d = depth_chunk * 64 + lane;
    const bool valid = d < n_depth;
    const int dc = valid ? d : n_depth - 1;
    const int ia = (int)i0;
    const int64_t pa = max(i0 - kNarrowReach + 1, (int64_t)0);
    const int64_t pb = min(i0 + F - 1 + kNarrowReach, n_nu);
    const int la = __builtin_amdgcn_readfirstlane(w.cnt_ge[pb + 1]);
    const int lb = __builtin_amdgcn_readfirstlane(w.cnt_ge[pa]);
    float nih[F], nil[F];
    double acc[F];
    bool act[F];
#pragma unroll
    for (int k = 0; k < F; ++k) {
        act[k] = i0 + k >= nu_begin && i0 + k < nu_begin + nu_count;
        const double nu = nus[min(i0 + k, n_nu - 1)];
        nih[k] = (float)nu;
        nil[k] = (float)(nu - (double)nih[k]);
        acc[k] = 0.0;
    }
    // (chunks of 64 candidates aligned to the LIST — base = 64 k — not to this frequency's first candidate: the fp32 sum of a chunk
    // is folded into the fp64 sum once per chunk, and which terms share a chunk must not depend on where the candidate range of a
    // frequency, or of a group of F frequencies, happens to start — F follows the launch's width, a shard's differs from the whole grid's)
    for (int base = la & ~63; base < lb; base += 64) {
        const int lc = base + lane;
        bool rel = false;
        int c = 0;
        if (lc >= la && lc < lb) {
            const int hwm = w.nhw_max[lc];
            c = w.centre[lc];
            rel = hwm > 0 && ia + (F - 1) >= c - hwm && ia < c + hwm;
        }
        unsigned long long m = __ballot(rel);
        float acc32[F];
#pragma unroll
        for (int k = 0; k < F; ++k) acc32[k] = 0.f;
        int h = 0, cl = 0;
        float y = 0.f, amp = 0.f, inv = 0.f;
        float2v lnu = {0.f, 0.f};
        if (m) {
            const int bit = __builtin_ctzll(m);
            const int l = base + bit;
            const size_t o = (size_t)l * n_depth + dc;
            cl = __builtin_amdgcn_readlane(c, bit);
            h = w.nhw[o], y = w.n_y32[o], amp = w.n_amp32[o], inv = w.n_inv32[o], lnu = w.lnu32[l];
        }
        while (m) {
            m &= m - 1;
            int h_n = 0, cl_n = 0;
            float y_n = 0.f, amp_n = 0.f, inv_n = 0.f;
            float2v lnu_n = {0.f, 0.f};
            if (m) {
                const int bit = __builtin_ctzll(m);
                const int l = base + bit;
                const size_t o = (size_t)l * n_depth + dc;
                cl_n = __builtin_amdgcn_readlane(c, bit);
                h_n = w.nhw[o], y_n = w.n_y32[o], amp_n = w.n_amp32[o], inv_n = w.n_inv32[o], lnu_n = w.lnu32[l];
            }
            const int lo = cl - h, hi = cl + h;
            if (valid && ia + (F - 1) >= lo && ia < hi) {
#pragma unroll
                for (int k = 0; k < F; ++k)
                    if (act[k] && ia + k >= lo && ia + k < hi) acc32[k] = voigt_add32(acc32[k], ((nih[k] - lnu.x) + (nil[k] - lnu.y)) * inv, y, amp);
            }
            h = h_n, cl = cl_n, y = y_n, amp = amp_n, inv = inv_n, lnu = lnu_n;
        }
#pragma unroll
        for (int k = 0; k < F; ++k) acc[k] += (double)acc32[k];
    }
#pragma unroll
    for (int k = 0; k < F; ++k)
        if (valid && act[k]) plane[(size_t)d * pld + (i0 + k - nu_begin)] = acc[k];
}

// ... and the subset walk of dense long lists (line_narrow_subsets) in the mixed-precision mode: the fp32 terms of one trip — the 64
// candidates a wave takes out of a 256-line stretch of the list — gather in an fp32 sum that goes into the fp64 sum once per trip.
template <int F>
__device__ __forceinline__ void line_narrow_subsets32(const int64_t i0, const int depth_chunk, const int subset, int n_depth, int64_t n_nu,
                                                      const double* __restrict__ nus, int64_t nu_begin, int64_t nu_count, LineWork w,
                                                      double* __restrict__ acc_out)
{
    const int lane = threadIdx.x & 63;
    const int d = depth_chunk * 64 + lane;
    const bool valid = d < n_depth;
    const int dc = valid ? d : n_depth - 1;
    const int ia = (int)i0;
    const int64_t pa = max(i0 - kNarrowReach + 1, (int64_t)0);
    const int64_t pb = min(i0 + F - 1 + kNarrowReach, n_nu);
    const int la = __builtin_amdgcn_readfirstlane(w.cnt_ge[pb + 1]);
    const int lb = __builtin_amdgcn_readfirstlane(w.cnt_ge[pa]);
    float nih[F], nil[F];
    double acc[F];
    bool act[F];
#pragma unroll
    for (int k = 0; k < F; ++k) {
        act[k] = i0 + k >= nu_begin && i0 + k < nu_begin + nu_count;
        const double nu = nus[min(i0 + k, n_nu - 1)];
        nih[k] = (float)nu;
        nil[k] = (float)(nu - (double)nih[k]);
        acc[k] = 0.0;
    }
    const int lane_off = ((lane >> 4) << 6) + (subset << 4) + (lane & 15);
    for (int base = la & ~255; base < lb; base += 256) {
        const int lc = base + lane_off;
        bool rel = false;
        int c = 0;
        if (lc >= la && lc < lb) {
            const int hwm = w.nhw_max[lc];
            c = w.centre[lc];
            rel = hwm > 0 && ia + (F - 1) >= c - hwm && ia < c + hwm;
        }
        unsigned long long m = __ballot(rel);
        float acc32[F];
#pragma unroll
        for (int k = 0; k < F; ++k) acc32[k] = 0.f;
        int h = 0, cl = 0;
        float y = 0.f, amp = 0.f, inv = 0.f;
        float2v lnu = {0.f, 0.f};
        if (m) {
            const int bit = __builtin_ctzll(m);
            const int l = base + ((bit >> 4) << 6) + (subset << 4) + (bit & 15);
            const size_t o = (size_t)l * n_depth + dc;
            cl = __builtin_amdgcn_readlane(c, bit);
            h = w.nhw[o], y = w.n_y32[o], amp = w.n_amp32[o], inv = w.n_inv32[o], lnu = w.lnu32[l];
        }
        while (m) {
            m &= m - 1;
            int h_n = 0, cl_n = 0;
            float y_n = 0.f, amp_n = 0.f, inv_n = 0.f;
            float2v lnu_n = {0.f, 0.f};
            if (m) {
                const int bit = __builtin_ctzll(m);
                const int l = base + ((bit >> 4) << 6) + (subset << 4) + (bit & 15);
                const size_t o = (size_t)l * n_depth + dc;
                cl_n = __builtin_amdgcn_readlane(c, bit);
                h_n = w.nhw[o], y_n = w.n_y32[o], amp_n = w.n_amp32[o], inv_n = w.n_inv32[o], lnu_n = w.lnu32[l];
            }
            const int lo = cl - h, hi = cl + h;
            if (valid && ia + (F - 1) >= lo && ia < hi) {
#pragma unroll
                for (int k = 0; k < F; ++k)
                    if (act[k] && ia + k >= lo && ia + k < hi) acc32[k] = voigt_add32(acc32[k], ((nih[k] - lnu.x) + (nil[k] - lnu.y)) * inv, y, amp);
            }
            h = h_n, cl = cl_n, y = y_n, amp = amp_n, inv = inv_n, lnu = lnu_n;
        }
#pragma unroll
        for (int k = 0; k < F; ++k) acc[k] += (double)acc32[k];
    }
#pragma unroll
    for (int k = 0; k < F; ++k) acc_out[k] = acc[k];  // (the four subsets of a group meet in the kernel's common reduction)
}

// FAR FIELD of the line opacity (far_eligible above): the third plane of the line kernels.  One workgroup owns (depth d, a unit of
// 4 RF consecutive global tiles, RF = 1 or 2); lane <-> (tile 4 r + lane / 16 of the unit, Chebyshev node lane % 16), r < RF.  Its
// n_split waves (the line kernel's, whose far role this is; kFarSplit in a launch of its own) go through the candidate lists exactly
// as the wide role does (hlist, then the wlist range around the unit; chunk q of 64 candidates belongs to wave q mod n_split): each
// lane tests ONE candidate against the unit (its whole span first, the single tiles where that does not settle it) and keeps the
// mask of the tiles it is far from; the hits are queued in LDS in list order and evaluated 64 at a time — records fetched by the
// lanes in one round trip — at the lanes' nodes with the wide role's own arithmetic (x = fma(dnu, inv, c0), region1_add), test-free
// when the candidate is far from every tile of the unit.  The waves' node sums meet in LDS (subset order), wave 0 turns the 16 node
// values of a tile into the coefficients of its Chebyshev series (kFarCoef), and every wave evaluates the series of its share of
// the tiles at their grid points (Clenshaw) and writes the plane — zeros where a tile has no far field.  The sum of a (tile, node)
// adds its lines in list order within a subset and the subsets in order, whatever RF and whatever the shard: RF is pure scheduling.
// F32 (the fp32-mixed tolerance mode, units of 8 tiles): the node sums in PACKED fp32 — a lane's two nodes are one float pair, a hit is
// region1_f32x2 on the pre-pass's 32-byte fp32 record (nine instructions where the fp64 form takes twenty-three for the pair) — and the
// Chebyshev transform and the Clenshaw recurrence in fp64 as before.  Per term ~1e-6 relative (the node's distance from the line is
// known to 3e-7: offsets from an fp32 base frequency, exact difference of two fp32 frequencies), well inside the mode's 1e-4.
template <int R, int RF, bool F32 = false>
__device__ __forceinline__ void line_far_body(const int block, const int units, const int n_split, int n_depth, int64_t n_nu, const double* __restrict__ nus,
                                              int64_t nu_begin, int64_t nu_count, int64_t n_lines, LineWork w, double* __restrict__ plane, int64_t pld,
                                              double* __restrict__ s_far)
{
    // s_far: [n_split][64 RF] partial node sums, [64 RF] node values, [64 RF] coefficients, [n_split][kFarWaveLdsDoubles]
    constexpr int kTile = 64 * R, kUnitTiles = 4 * RF;
    const int lane = threadIdx.x & 63;
    const int split = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int d = block / units, u = block - d * units;
    const int64_t T_first = nu_begin / kTile, T_last = (nu_begin + nu_count - 1) / kTile;  // the global tiles that hold the shard's columns
    const int64_t Tu = (T_first / kUnitTiles + u) * kUnitTiles;                          // units are aligned to the global tiles too
    const int64_t Ta = max(Tu, T_first), Tb = min(Tu + kUnitTiles - 1, T_last);
    if (Ta > Tb) return;
    const int64_t g0 = Ta * kTile, g1 = min((Tb + 1) * kTile, n_nu);  // the grid points of the tiles this workgroup handles
    const int64_t s0 = nu_begin, s1 = nu_begin + nu_count;

    // the tiles' far ranges, wave-uniform (scalar loads): the candidate tests below
    static_assert(kTile == kFarTile, "the far field lives on 256-point tiles");
    int ihi[kUnitTiles], ilo[kUnitTiles];
    unsigned all_mask = 0;
#pragma unroll
    for (int t = 0; t < kUnitTiles; ++t) {
        const int64_t T = Tu + t;
        const bool mine = T >= Ta && T <= Tb;
        ihi[t] = mine ? w.far_range[2 * T] : 0;
        ilo[t] = mine ? w.far_range[2 * T + 1] : 0x7FFFFFFF;
        all_mask |= (ihi[t] != 0 || ilo[t] != 0x7FFFFFFF) ? (1u << t) : 0u;  // (0, INT_MAX): no far field
    }
    // ... and what decides most candidates without a look at the single tiles: the span of the tiles that have a far field (they are
    // consecutive: only the grid's last tile can lack one) and the tightest of their ranges
    int span_lo = 0x7FFFFFFF, span_hi = 0, min_ihi = 0x7FFFFFFF, max_ilo = 0;
#pragma unroll
    for (int t = 0; t < kUnitTiles; ++t) {
        if ((all_mask >> t) & 1u) {
            span_lo = min(span_lo, (int)((Tu + t) * kTile));
            span_hi = max(span_hi, (int)((Tu + t + 1) * kTile));
            min_ihi = min(min_ihi, ihi[t]);
            max_ilo = max(max_ilo, ilo[t]);
        }
    }
    if (all_mask == 0) {  // nothing but a partial last tile: zeros
        for (int t = split; t < kUnitTiles; t += n_split) {
            const int64_t T = Tu + t;
            if (T < Ta || T > Tb) continue;
            for (int p = 0; p < R; ++p) {
                const int64_t i = T * kTile + lane + 64 * p;
                if (i < n_nu && i >= s0 && i < s1) plane[(size_t)d * pld + (i - s0)] = 0.0;
            }
        }
        return;
    }
    // the lane's nodes as offsets from a base frequency (the wide role's form of x) that depends on neither the shard nor RF: the
    // first frequency of the block of 16 global tiles the unit lies in
    const double nu_base_v = nus[(Tu / 16) * 16 * kTile];
    const double nu_base = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(nu_base_v)), __builtin_amdgcn_readfirstlane(__double2loint(nu_base_v)));
    double nu_base_vec = nu_base;
    asm("" : "+v"(nu_base_vec));
    double dnu[RF], acc[RF];
    const int tg = lane >> 4;
    const double node = kFarNode[lane & 15];
#pragma unroll
    for (int r = 0; r < RF; ++r) {
        const int64_t T = Tu + 4 * r + tg, i0 = T * kTile;
        acc[r] = 0.0;
        dnu[r] = 0.0;
        if (T >= Ta && T <= Tb && i0 + kTile <= n_nu) {
            double c, h;
            far_tile_geometry(nus[i0], nus[i0 + kTile - 1], c, h);
            dnu[r] = fma(h, node, c - nu_base);
        }
    }
    const size_t row = (size_t)d * n_lines;
    const WideScan* __restrict__ scan_row = w.wscan + row;
    const WideScan* __restrict__ hscan_row = w.hscan + row;
    const WideRec* __restrict__ rec_row = w.wrec + row;
    [[maybe_unused]] const WideRec32* __restrict__ rec32_row = F32 ? w.wrec32 + row : nullptr;
    const int n_h = w.hlist ? __builtin_amdgcn_readfirstlane(*w.hcount) : 0;
    // F32: the lane's two nodes as fp32 offsets from the base frequency ROUNDED to fp32 (what the fp32 records' hi parts are differences to)
    static_assert(!F32 || RF == 2, "the packed fp32 far role pairs a lane's two nodes");
    [[maybe_unused]] float base_h = (float)nu_base;
    [[maybe_unused]] float2v dq32 = {0.f, 0.f}, acc32 = {0.f, 0.f};
    if constexpr (F32) {
        asm("" : "+v"(base_h));
        const double shift = nu_base - (double)base_h;
        dq32.x = (float)(dnu[0] + shift);
        dq32.y = (float)(dnu[RF - 1] + shift);
    }

    // Scan and evaluation are decoupled: a wave that fetched the record of every hit when it met it spent its time waiting for one
    // dependent load after the other (~500 hits, a microsecond each).  The hits of the chunks are QUEUED in LDS instead — line index
    // and tile mask, in list order — and a full queue (64 hits) is evaluated at once: lane j fetches hit j's 48-byte record, all 64
    // in ONE round trip, into LDS, and the wave goes through them with broadcast reads.
    double* const wave_lds = s_far + (size_t)(n_split + 2) * RF * 64 + (size_t)split * kFarWaveLdsDoubles;
    WideRec* const stage = reinterpret_cast<WideRec*>(wave_lds);  // 64 x 48 B
    [[maybe_unused]] WideRec32* const stage32 = reinterpret_cast<WideRec32*>(wave_lds);  // (F32: 64 x 32 B in the same space)
    int* const q_line = reinterpret_cast<int*>(wave_lds + 64 * 6);
    int* const q_mask = q_line + 64;
    int count = 0;
    auto flush = [&](int cnt) {
        if (cnt == 0) return;
        wave_sync();
        const bool mine = lane < cnt;
        const unsigned qm = mine ? (unsigned)q_mask[lane] : 0u;
        if constexpr (F32) {
            if (mine) stage32[lane] = rec32_row[q_line[lane]];
        } else {
            if (mine) stage[lane] = rec_row[q_line[lane]];
        }
        const unsigned long long mfull = __ballot(mine && qm == all_mask);
        wave_sync();
        if constexpr (F32) {
            for (int k = 0; k < cnt; ++k) {
                const WideRec32 cur = stage32[k];  // (one address for the whole wave: a broadcast read)
                const float c0 = region1_c0(base_h, cur);
                if ((mfull >> k) & 1) {  // far from every tile of the unit: no per-lane tests
                    acc32 = region1_f32x2(acc32, dq32, c0, cur);
                } else {
                    const unsigned tm = (unsigned)__builtin_amdgcn_readlane((int)qm, k);
                    // (the accumulating FMA of the test-free form, kept or dropped per node: a tile's sum must not depend on whether its
                    // unit's other tiles — which a shard may not own — let the hit take the test-free path)
                    const float2v v = region1_f32x2(acc32, dq32, c0, cur);
                    acc32.x = ((tm >> tg) & 1u) ? v.x : acc32.x;
                    acc32.y = ((tm >> (4 + tg)) & 1u) ? v.y : acc32.y;
                }
            }
            wave_sync();
            return;
        }
        for (int k = 0; k < cnt; ++k) {
            const WideRec cur = stage[k];  // (one address for the whole wave: a broadcast read)
            const RegionI k1 = {cur.yk, cur.cv, cur.cd};
            const double c0 = (nu_base_vec - cur.lnu) * cur.inv;
            if (((mfull >> k) & 3) == 3 && k + 1 < cnt) {
                // two test-free hits at once: two independent chains of dependent instructions (a shard's launch has two or three
                // waves per SIMD: nothing else fills the gaps); the sums keep list order
                const WideRec nxt = stage[k + 1];
                const RegionI k2 = {nxt.yk, nxt.cv, nxt.cd};
                const double c1 = (nu_base_vec - nxt.lnu) * nxt.inv;
#pragma unroll
                for (int r = 0; r < RF; ++r) {
                    acc[r] = region1_add(acc[r], fma(dnu[r], cur.inv, c0), k1);
                    acc[r] = region1_add(acc[r], fma(dnu[r], nxt.inv, c1), k2);
                }
                ++k;
            } else if ((mfull >> k) & 1) {  // far from every tile of the unit: no per-lane tests
#pragma unroll
                for (int r = 0; r < RF; ++r) acc[r] = region1_add(acc[r], fma(dnu[r], cur.inv, c0), k1);
            } else {
                const unsigned tm = (unsigned)__builtin_amdgcn_readlane((int)qm, k);
#pragma unroll
                for (int r = 0; r < RF; ++r) {
                    if (((tm >> (4 * r)) & 15u) == 0) continue;  // (scalar)
                    const bool take = (tm >> (4 * r + tg)) & 1u;
                    acc[r] = region1_add_if(acc[r], fma(dnu[r], cur.inv, c0), k1, take);
                }
            }
        }
        wave_sync();  // (the queue and the staged records are written again)
    };

    for (int pass = w.hlist ? 0 : 1; pass < 2; ++pass) {
        int ka = 0, kb = n_h;
        if (pass == 1) {
            kb = (int)n_lines;
            if (w.hlist) {  // (the wide role's candidate range, around the unit instead of a tile)
                const int64_t pa = max(g0 - kMediumHalfWidth + 1, (int64_t)0), pb = min(g1 + kMediumHalfWidth - 1, n_nu);
                ka = __builtin_amdgcn_readfirstlane(w.wrank[w.cnt_ge[pb + 1]]);
                kb = __builtin_amdgcn_readfirstlane(w.wrank[w.cnt_ge[pa]]);
                if (w.sel) {
                    ka = max(ka, __builtin_amdgcn_readfirstlane(w.wrank[w.sel[0]]));
                    kb = min(kb, __builtin_amdgcn_readfirstlane(w.wrank[w.sel[1]]));
                }
            }
        }
        if (kb <= ka) continue;
        const int q_first = ka >> 6, q_last = (kb - 1) >> 6;
        int q = q_first + ((split - q_first % n_split) + n_split) % n_split;
        auto fetch = [&](int qq, int& line, WideScan& sc) {
            const int k = qq * 64 + lane;
            line = -1;
            sc = WideScan{0, 0, 0, 0};
            if (qq <= q_last && k >= ka && k < kb) {
                if (pass == 0) {
                    line = w.hlist[k];
                    sc = w.hscan ? hscan_row[k] : scan_row[line];
                } else {
                    line = w.hlist ? w.wlist[k] : k;
                    sc = scan_row[line];
                }
            }
        };
        int line_next;
        WideScan sc_next;
        fetch(q, line_next, sc_next);
        for (; q <= q_last; q += n_split) {
            const int line = line_next;
            const WideScan sc = sc_next;
            fetch(q + n_split, line_next, sc_next);
            // Most candidates are far from every tile of the unit (a strong line somewhere else on the grid) or reach none of them: one test
            // of the whole span each — far_eligible of the span with the tightest range implies it of every tile — and the tile-by-tile
            // tests only for a chunk that holds a candidate neither settles
            const bool all_far = far_eligible(sc, span_lo, span_hi, min_ihi, max_ilo);
            const bool none = (line < 0) | (sc.hi - sc.lo < kTile) | (sc.hi <= span_lo) | (sc.lo >= span_hi);
            unsigned tmask = all_far && !none ? all_mask : 0u;
            if (__ballot(!all_far && !none)) {
                unsigned tm = 0;
#pragma unroll
                for (int t = 0; t < kUnitTiles; ++t) {
                    const int i0 = (int)min((Tu + t) * kTile, n_nu);  // (tiles beyond the grid: never far)
                    tm |= far_eligible(sc, i0, i0 + kTile, ihi[t], ilo[t]) ? (1u << t) : 0u;
                }
                if (!all_far && !none) tmask = tm;
            }
            // the hits join the wave's queue (list order); a full queue is evaluated (flush)
            const unsigned long long m = __ballot(tmask != 0);
            const int n = __popcll(m);
            if (n == 0) continue;
            if (count + n > 64) {
                flush(count);
                count = 0;
            }
            if (tmask != 0) {
                const int at = count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                q_line[at] = line;
                q_mask[at] = (int)tmask;
            }
            count += n;
        }
    }
    flush(count);
    if constexpr (F32) acc[0] = (double)acc32.x, acc[RF - 1] = (double)acc32.y;

    // the subsets' node sums, in subset order
    double* red = s_far;
    double* sV = s_far + (size_t)n_split * RF * 64;
    double* sC = sV + RF * 64;
    if (split > 0) {
#pragma unroll
        for (int r = 0; r < RF; ++r) red[((size_t)split * RF + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (split == 0) {
        for (int s = 1; s < n_split; ++s) {
#pragma unroll
            for (int r = 0; r < RF; ++r) acc[r] = add_rn(acc[r], red[((size_t)s * RF + r) * 64 + lane]);
        }
#pragma unroll
        for (int r = 0; r < RF; ++r) sV[r * 64 + lane] = acc[r];
        wave_sync();
        // node values -> Chebyshev coefficients: lane <-> (tile, k)
        const int k = lane & 15;
#pragma unroll
        for (int r = 0; r < RF; ++r) {
            const double* v = sV + r * 64 + (lane & 48);
            double c = 0.0;
#pragma unroll
            for (int j = 0; j < kFarNodes; ++j) c = fma(kFarCoef[k][j], v[j], c);
            sC[r * 64 + lane] = c;  // = sC[(4 r + tg) * 16 + k]
        }
    }
    __syncthreads();
    // the series at the grid points of the tiles (Clenshaw), one tile at a time per wave
    for (int t = split; t < kUnitTiles; t += n_split) {
        const int64_t T = Tu + t, i0 = T * kTile;
        if (T < Ta || T > Tb) continue;
        bool okt = i0 + kTile <= n_nu;
        double c = 0.0, h = 1.0;
        if (okt) {
            far_tile_geometry(nus[i0], nus[i0 + kTile - 1], c, h);
            okt = h > 0.0;
        }
        const double rh = okt ? 1.0 / h : 0.0;
        const double* cf = sC + t * kFarNodes;
#pragma unroll
        for (int p = 0; p < R; ++p) {
            const int64_t i = i0 + lane + 64 * p;
            if (i >= n_nu) continue;
            double val = 0.0;
            if (okt) {
                const double x = (nus[i] - c) * rh, x2 = x + x;
                double b1 = 0.0, b2 = 0.0;
#pragma unroll
                for (int k = kFarNodes - 1; k >= 1; --k) {
                    const double b0 = fma(x2, b1, cf[k] - b2);
                    b2 = b1, b1 = b0;
                }
                val = fma(x, b1, cf[0] - b2);
            }
            if (i >= s0 && i < s1) plane[(size_t)d * pld + (i - s0)] = val;
        }
    }
}

// (the far field as a launch of its own: experiment knob SDX_FAR_LAUNCH; by default its workgroups are the FIRST of the line kernel's grid)
template <int R, int RF>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_line_far(int units, int n_split, int n_depth, int64_t n_nu,
                                                                                           const double* __restrict__ nus, int64_t nu_begin, int64_t nu_count,
                                                                                           int64_t n_lines, LineWork w, double* __restrict__ plane, int64_t pld)
{
    extern __shared__ double s_far[];
    line_far_body<R, RF>(blockIdx.x, units, n_split, n_depth, n_nu, nus, nu_begin, nu_count, n_lines, w, plane, pld, s_far);
}

// Both line kernels in ONE launch of workgroups of S waves (S = number of line subsets): workgroups [0, n_wide) take the
// wide role — one (depth, tile) each, wave s walks subset s — depth slowest, hottest layers first; the rest take the narrow
// role, one frequency per wave.  The two roles only share the pre-pass, and each leaves issue slots idle on its own; a
// cross-stream fork/join would cost two ~12 us inter-queue edges per step, one grid costs nothing.
// roles: bit 0 wide, bit 1 narrow (both by default; one at a time for split-launch profiling, SDX_SPLIT_LAUNCHES=1); bits 16-17: the
// far field's workgroups (line_far_body) come FIRST in the grid, n_depth x far_units of them (fp64 kernels with a far field).
// Output planes: [0] the wide windows (all subsets summed), [1] the narrow windows.
template <int R, bool MIXED, bool SUBSETS, bool FAR = false>
__device__ __forceinline__ void line_all_body(int n_wide, int tiles, int n_split, int n_depth, int64_t n_nu,
                                                   const double* __restrict__ nus, int64_t nu_begin, int64_t nu_count,
                                                   int64_t n_lines, const double* __restrict__ line_nus, LineWork w,
                                                   double* __restrict__ planes, int64_t pld, int roles, int far_units)
{
    extern __shared__ double s_wide[];  // n_split x kWideLdsDoubles
    int b = blockIdx.x;
    // (fp32-mixed kernels with a far field: what is left to the wide role — a line's near zone, window edges, kept cores — is walked in
    // fp64 by the queued walk like the fp64 kernels'; the narrow role and the formal solution stay the mode's fp32)
    constexpr bool WM = MIXED && !FAR;  // the wide role's arithmetic
    if constexpr (FAR) {
        // far role (roles bits 16-17: RF, 0 = the far field has a launch of its own): the FIRST workgroups of the grid — their waves are
        // the longest chains of the launch (a unit walks every huge line of the list) — n_depth x far_units of them
        const int rf = (roles >> 16) & 3;
        if (rf) {
            const int n_far = n_depth * far_units;
            if (b < n_far) {
                double* const far_plane = planes + (size_t)2 * n_depth * pld;
                // (fp32-mixed mode: the node sums in packed fp32 — units of 8 tiles, the default; the experiment knob's units of 4 stay fp64)
                if (rf == 2) line_far_body<R, 2, MIXED>(b, far_units, n_split, n_depth, n_nu, nus, nu_begin, nu_count, n_lines, w, far_plane, pld, s_wide);
                else line_far_body<R, 1>(b, far_units, n_split, n_depth, n_nu, nus, nu_begin, nu_count, n_lines, w, far_plane, pld, s_wide);
                return;
            }
            b -= n_far;
        }
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform, which the compiler cannot see: chunk and frequency indices stay scalar
    // SUBSETS (the kernel of dense long lists): both roles end in ONE reduction — every wave of a workgroup brings R partial sums per
    // lane (a wide wave: R points of its tile; a narrow wave: the R = F frequencies of the workgroup's group), they meet in LDS
    // and wave 0 adds them in subset order and writes.  One barrier in the kernel: with a second one in the narrow branch the
    // compiler spilled the WIDE walk's registers, and a kernel that touches scratch at all ran a third slower (round 5).
    [[maybe_unused]] double part[R];
    [[maybe_unused]] int out_row = 0, out_col = 0;      // where lane's sums go: row (depth), first column; columns step 64 (wide) or 1 (narrow)
    [[maybe_unused]] bool out_wide = true, out_valid = true;
    if (b < n_wide) {
        if (!(roles & 1)) return;
        // XCD-aware tile order: workgroup i runs on XCD i % 8, each with its own L2.  Within a depth the workgroups of one XCD
        // take CONTIGUOUS tiles (position p -> tile prefix(p % 8) + p / 8), so neighbouring tiles, whose line ranges
        // overlap, hit the same L2 instead of pulling the same records into all eight.
        const int wg = (roles >> 4) & 15;  // 0: one contiguous eighth of the tiles per XCD; g > 0: groups of g tiles going round the XCDs
        int tile, d;
        if (wg == 0) {
            const int p = b % tiles;
            d = b / tiles;
            tile = p >> 3;
            for (int f = 0; f < (p & 7); ++f) tile += (tiles - f + 7) >> 3;
        } else {
            // (the host pads the tiles of a depth to whole rounds of 8 g workgroups: b % 8 is then the XCD within every depth)
            const int tiles_pad = (tiles + 8 * wg - 1) / (8 * wg) * (8 * wg);
            const int p = b % tiles_pad, j = p >> 3;
            d = b / tiles_pad;
            tile = ((j / wg) * 8 + (p & 7)) * wg + j % wg;
            if (tile >= tiles) return;
        }
        if constexpr (SUBSETS) {
            line_wide_walk<R, WM, true, WM, FAR>(tile, wave, n_split, d, n_nu, nus, nu_begin, nu_count, n_lines, w, planes, pld, s_wide, part);
            out_row = d;
            out_col = (int)((nu_begin / (64 * R) + (int64_t)tile) * (64 * R)) + (int)(threadIdx.x & 63);
        } else {
            line_wide_walk<R, WM, false, false, FAR>(tile, wave, n_split, d, n_nu, nus, nu_begin, nu_count, n_lines, w, planes, pld, s_wide);
        }
    } else {
        if (!(roles & 2)) return;
        // A wave writes one value into each of the N_d rows of the narrow plane: the waves that fill a 64-byte sector of a
        // row (8 consecutive frequencies) should share an L2, or every XCD writes its own fragment of every sector back on
        // its own.  Workgroups p, p + 8, ... (one XCD) therefore take GROUPS of kNarrowGroup consecutive workgroups' worth
        // of frequencies, the groups going round the XCDs.  (Giving each XCD one contiguous eighth of the grid was measured
        // 45 % slower: the lines per grid point follow the frequency, so one XCD gets several times the work of another.)
        // (SUBSETS: a workgroup is ONE group of frequencies, not four — sixteen workgroups keep the 64 consecutive frequencies per XCD
        // whose lines' records then meet in one L2)
        constexpr int kNarrowGroup = SUBSETS ? 16 : 4;
        // F consecutive frequencies per wave (roles bits 8-11: 1, 2 or 4 — 8 was measured slower), groups aligned
        // to the global grid
        const int F = max(1, (roles >> 8) & 15);
        // dense long lists (a kernel of their own — SUBSETS — so that neither walk pays for the other's registers: with both narrow
        // walks in one kernel the WIDE role spilled, and a kernel that touches scratch memory ran a third slower): the workgroup's
        // four waves share one group of F = 4 frequencies
        constexpr bool subsets = SUBSETS;
        const int64_t g0 = nu_begin / F;
        const int64_t n_grp = (nu_begin + nu_count + F - 1) / F - g0;
        const int64_t n_narrow = n_grp * ((n_depth + 63) / 64);
        const int64_t n_nb = subsets ? n_narrow : (n_narrow + n_split - 1) / n_split;
        const int64_t p = b - n_wide, j = p >> 3;
        const int order = (roles >> 2) & 3;  // analysis knob (SDX_NARROW_ORDER): 0 grouped (default), 1 plain, 2 one block per XCD
        int64_t wg = ((j / kNarrowGroup) * 8 + (p & 7)) * kNarrowGroup + j % kNarrowGroup;
        if (order == 1) wg = p;
        if (order == 2) wg = (p & 7) * ((n_nb + 7) / 8) + j;
        if ((order == 2 && j >= (n_nb + 7) / 8) || wg >= n_nb) return;
        const int64_t c = subsets ? wg : wg * n_split + wave;
        if (c >= n_narrow) return;  // (subsets: the whole workgroup)
        const int64_t i0 = (g0 + c % n_grp) * F;
        const int chunk = (int)(c / n_grp);
        double* __restrict__ nplane = planes + (size_t)n_depth * pld;
        if constexpr (SUBSETS) {
            static_assert(!SUBSETS || R == 4, "the common reduction: R points per wide lane = F frequencies per narrow group");
            if constexpr (MIXED) line_narrow_subsets32<4>(i0, chunk, wave, n_depth, n_nu, nus, nu_begin, nu_count, w, part);
            else line_narrow_subsets<4>(i0, chunk, wave, n_depth, n_nu, nus, nu_begin, nu_count, line_nus, w, part);
            out_wide = false;
            out_row = chunk * 64 + (int)(threadIdx.x & 63);
            out_valid = out_row < n_depth;
            out_col = (int)i0;
        } else if constexpr (MIXED) {
            if (F == 4) line_narrow_group32<4>(i0, chunk, n_depth, n_nu, nus, nu_begin, nu_count, w, nplane, pld);
            else if (F == 2) line_narrow_group32<2>(i0, chunk, n_depth, n_nu, nus, nu_begin, nu_count, w, nplane, pld);
            else line_narrow_wave32(i0, chunk, n_depth, n_nu, nus, nu_begin, nu_count, w, nplane, pld);
        } else {
            if (F == 4) line_narrow_group<4>(i0, chunk, n_depth, n_nu, nus, nu_begin, nu_count, n_lines, line_nus, w, nplane, pld);
            else if (F == 2) line_narrow_group<2>(i0, chunk, n_depth, n_nu, nus, nu_begin, nu_count, n_lines, line_nus, w, nplane, pld);
            else line_narrow_wave(i0, chunk, n_depth, n_nu, nus, nu_begin, nu_count, n_lines, line_nus, w, nplane, pld);
        }
    }
    if constexpr (SUBSETS) {
        constexpr int kLdsStride = (FAR && SDX_WIDE_QUEUED) ? kWideFarLdsDoubles : kWideLdsDoubles;  // (the wide walk's)
        const int lane = threadIdx.x & 63;
        if (wave > 0) {
            double* mine = s_wide + (size_t)wave * kLdsStride;
#pragma unroll
            for (int r = 0; r < R; ++r) mine[r * 64 + lane] = part[r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll 1
            for (int s = 1; s < n_split; ++s) {
                const double* other = s_wide + (size_t)s * kLdsStride;
#pragma unroll
                for (int r = 0; r < R; ++r) part[r] = add_rn(part[r], other[r * 64 + lane]);
            }
            const int64_t s0 = nu_begin, s1 = nu_begin + nu_count;  // the columns this launch stores
            double* __restrict__ dst = planes + (out_wide ? (size_t)0 : (size_t)n_depth * pld) + (size_t)out_row * pld;
            const int stride = out_wide ? 64 : 1;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t col = (int64_t)out_col + (int64_t)r * stride;
                if (out_valid && col >= s0 && col < s1) dst[col - s0] = part[r];
            }
        }
    }
}

template <int R, bool SUBSETS = false, bool FAR = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(FAR ? SDX_FAR_WAVES : 7, 8))) void k_line_all(int n_wide, int tiles, int n_split, int n_depth, int64_t n_nu,
                                                   const double* __restrict__ nus, int64_t nu_begin, int64_t nu_count,
                                                   int64_t n_lines, const double* __restrict__ line_nus, LineWork w,
                                                   double* __restrict__ planes, int64_t pld, int roles, int far_units)
{
    line_all_body<R, false, SUBSETS, FAR>(n_wide, tiles, n_split, n_depth, n_nu, nus, nu_begin, nu_count, n_lines, line_nus, w, planes, pld, roles, far_units);
}
// the mixed-precision variant: 512-point tiles; the register budget is capped at 128 (4 waves per SIMD) — what exceeds it
// sits in the rarely taken fp64 general path
template <int R, bool SUBSETS = false, bool FAR = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(R == 4 ? 6 : 4, 8))) void k_line_all_mixed(
    int n_wide, int tiles, int n_split, int n_depth, int64_t n_nu, const double* __restrict__ nus, int64_t nu_begin, int64_t nu_count,
    int64_t n_lines, const double* __restrict__ line_nus, LineWork w, double* __restrict__ planes, int64_t pld, int roles, int far_units)
{
    line_all_body<R, true, SUBSETS, FAR>(n_wide, tiles, n_split, n_depth, n_nu, nus, nu_begin, nu_count, n_lines, line_nus, w, planes, pld, roles, far_units);
}

// out (+)= sum over the S line subsets, in subset order
__global__ __launch_bounds__(kBlock) void k_reduce_partials(int n_depth, int64_t nu_count, int n_split,
                                                            const double* __restrict__ partial, int64_t pld,
                                                            double* __restrict__ out, int64_t out_ld, int accumulate)
{
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int d = blockIdx.y;
    if (j >= nu_count) return;
    double v = partial[(size_t)d * pld + j];
    for (int s = 1; s < n_split; ++s) v = add_rn(v, partial[((size_t)s * n_depth + d) * pld + j]);
    double* p = out + (size_t)d * out_ld + j;
    *p = accumulate ? add_rn(*p, v) : v;
}

// ------------------------------------------------------------------------------------------------
// element-wise entry points
__global__ __launch_bounds__(kBlock) void k_faddeeva(int64_t n, const double* __restrict__ z, double* __restrict__ wout)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const c64 r = faddeeva_full(c64{z[2 * i], z[2 * i + 1]});
    wout[2 * i] = r.re;
    wout[2 * i + 1] = r.im;
}

__global__ __launch_bounds__(kBlock) void k_voigt_profile(int64_t n, const double* __restrict__ dnu,
                                                          const double* __restrict__ dw, const double* __restrict__ g,
                                                          double* __restrict__ phi)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) phi[i] = voigt_profile_full(dnu[i], dw[i], g[i]);
}

// The routine the line kernels evaluate per (line, depth, frequency) — region1_setup + voigt_term, i.e. the FMA /
// real-part-only variant of voigt.py:17-86,113-150 — exposed element-wise so that it can be pinned point by point against
// the reference's Faddeeva / Voigt golden vectors (the element-wise sdx_faddeeva_dev / sdx_voigt_profile_dev run the
// reference-order routine faddeeva_full instead).  out = amp * Re w((delta_nu + i gamma / (sqrt(pi) pi)) / doppler_width)
// with the pre-pass's derived constants: inv_dw = 1 / dw, y = (gamma / (sqrt(pi) pi)) / dw, amp as given.
__global__ __launch_bounds__(kBlock) void k_voigt_term(int64_t n, const double* __restrict__ dnu, const double* __restrict__ inv_dw,
                                                       const double* __restrict__ y, const double* __restrict__ amp,
                                                       double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const RegionI k1 = region1_setup(y[i], amp[i]);
    out[i] = voigt_term(dnu[i], inv_dw[i], y[i], amp[i], k1);
}

// The fp32 routine of the mixed-precision mode's narrow role (voigt_add32), element-wise: inputs rounded to fp32 as the
// pre-pass and the kernel round them.
__global__ __launch_bounds__(kBlock) void k_voigt_term32(int64_t n, const double* __restrict__ dnu, const double* __restrict__ inv_dw,
                                                         const double* __restrict__ y, const double* __restrict__ amp,
                                                         double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    out[i] = (double)voigt_add32(0.f, (float)dnu[i] * (float)inv_dw[i], (float)y[i], (float)amp[i]);
}

// F_lambda = F_nu * nu / lambda (stardis/base.py:137-141: spectrum_lambda; the unit conversion there has scale 1)
__global__ __launch_bounds__(kBlock) void k_flux_nu_to_lambda(int64_t n, const double* __restrict__ f_nu, const double* __restrict__ nus,
                                                              const double* __restrict__ lambdas, double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = mul_rn(f_nu[i], nus[i]) / lambdas[i];
}

__global__ __launch_bounds__(kBlock) void k_blackbody(int n_depth, int64_t n_nu, const double* __restrict__ nus,
                                                      const double* __restrict__ temps, double* __restrict__ out, int64_t ld)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int d = blockIdx.y;
    if (i < n_nu) out[(size_t)d * ld + i] = planck(nus[i], temps[d]);
}

__global__ __launch_bounds__(kBlock) void k_weights(int64_t n, const double* __restrict__ tau, double* __restrict__ w0,
                                                    double* __restrict__ w1, double* __restrict__ w2)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double a, b, c;
    rt_weights(tau[i], a, b, c);
    w0[i] = a;
    w1[i] = b;
    w2[i] = c;
}

// ------------------------------------------------------------------------------------------------
// broadening kernels (scalar formulas: sdx_broadening.h)
__global__ __launch_bounds__(kBlock) void k_calc_gamma(int64_t n_lines, int n_depth, const int* __restrict__ z,
                                                       const int* __restrict__ ion, const double* __restrict__ e_ion,
                                                       const double* __restrict__ e_up, const double* __restrict__ e_lo,
                                                       const double* __restrict__ a_ul, const double* __restrict__ ne,
                                                       const double* __restrict__ temps, const double* __restrict__ nh,
                                                       int flags, double* __restrict__ out)
{
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= n_lines * n_depth) return;
    const int64_t l = k / n_depth;
    const int d = (int)(k - l * n_depth);
    const double nu_ = n_effective(ion[l], e_ion[l], e_up[l]);
    const double nl_ = n_effective(ion[l], e_ion[l], e_lo[l]);
    const double g_lin = ((flags & 1) && z[l] == 1) ? gamma_linear_stark(nu_, nl_, ne[d]) : 0.0;
    const double g_q = (flags & 2) ? gamma_quadratic_stark(ion[l], nu_, nl_, ne[d], temps[d]) : 0.0;
    const double g_w = (flags & 4) ? gamma_van_der_waals(ion[l], nu_, nl_, temps[d], nh[d]) : 0.0;
    const double g_r = (flags & 8) ? a_ul[l] : 0.0;
    out[k] = add_rn(add_rn(add_rn(g_lin, g_q), g_w), g_r);  // broadening.py:649-654
}

__global__ __launch_bounds__(kBlock) void k_doppler_widths(int64_t n_lines, int n_depth, const double* __restrict__ lnu,
                                                           const double* __restrict__ mass, const double* __restrict__ temps,
                                                           double xi, double* __restrict__ out)
{
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= n_lines * n_depth) return;
    const int64_t l = k / n_depth;
    const int d = (int)(k - l * n_depth);
    out[k] = doppler_width(lnu[l], temps[d], mass[l], xi);
}

// the three dense tables of the reference from a LineParams description (parity checks of the f1 path; any output may be null)
__global__ __launch_bounds__(kBlock) void k_line_params(int64_t n_lines, int n_depth, const double* __restrict__ line_nus,
                                                        LineParams lp, double* __restrict__ alphas, double* __restrict__ gammas,
                                                        int gamma_cols, double* __restrict__ doppler)
{
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= n_lines * n_depth) return;
    const int64_t l = k / n_depth;
    const int d = (int)(k - l * n_depth);
    const double lnu = line_nus[l];
    const GenDepth D = gen_depth(lp, d);
    const GenLine L = gen_line(lp, lnu, l);
    if (alphas) alphas[k] = gen_alpha(lp, L, D, d, n_depth);
    if (doppler) doppler[k] = gen_doppler(lp, L, D);
    if (gammas && (gamma_cols > 1 || d == 0)) gammas[l * gamma_cols + (gamma_cols > 1 ? d : 0)] = gen_gamma(lp, L, D);
}

// plasma/base.py:130-175 AlphaLine: alpha = ((ALPHA_COEFFICIENT * n_lower) * stimulated_emission_factor) * f_lu, n_lower
// gathered from the level populations by lines_lower_level_index (numpy take, mode="raise": the host checks the range)
__global__ __launch_bounds__(kBlock) void k_alpha_line_levels(int64_t n_lines, int n_depth, const double* __restrict__ level_density,
                                                              const int* __restrict__ lower_index, const double* __restrict__ stim,
                                                              const double* __restrict__ f_lu, double coefficient,
                                                              double* __restrict__ alphas)
{
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= n_lines * n_depth) return;
    const int64_t l = k / n_depth;
    const int d = (int)(k - l * n_depth);
    const double n_lower = level_density[(size_t)lower_index[l] * n_depth + d];
    alphas[k] = mul_rn(mul_rn(mul_rn(coefficient, n_lower), stim[k]), f_lu[l]);
}

// the reference's element-wise ufuncs (broadening.py:69-71, :140-146, :232-234, :346-360, :476-490)
enum BroadeningOp { kOpDoppler = 0, kOpNEff = 1, kOpLinearStark = 2, kOpQuadraticStark = 3, kOpVanDerWaals = 4 };
__global__ __launch_bounds__(kBlock) void k_broadening_scalar(int op, int64_t n, const double* __restrict__ a,
                                                              const double* __restrict__ b, const double* __restrict__ c,
                                                              const double* __restrict__ d, const double* __restrict__ e,
                                                              double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double r = 0.0;
    switch (op) {
        case kOpDoppler: r = doppler_width(a[i], b[i], c[i], d[i]); break;                              // nu, T, mass, xi
        case kOpNEff: r = n_effective((int)a[i], b[i], c[i]); break;                                    // ion, E_ion, E_lev
        case kOpLinearStark: r = gamma_linear_stark(a[i], b[i], c[i]); break;                           // n_up, n_lo, n_e
        case kOpQuadraticStark: r = gamma_quadratic_stark((int)a[i], b[i], c[i], d[i], e[i]); break;    // ion, n_up, n_lo, n_e, T
        case kOpVanDerWaals: r = gamma_van_der_waals((int)a[i], b[i], c[i], d[i], e[i]); break;         // ion, n_up, n_lo, T, n_H
    }
    out[i] = r;
}

__global__ __launch_bounds__(kBlock) void k_calc_vald_gamma(int64_t n_lines, int n_depth, const int* __restrict__ z,
                                                            const int* __restrict__ ion, const double* __restrict__ e_ion,
                                                            const double* __restrict__ e_up, const double* __restrict__ e_lo,
                                                            const double* __restrict__ a_ul, const double* __restrict__ stark,
                                                            const double* __restrict__ waals, const double* __restrict__ mass,
                                                            const double* __restrict__ ne, const double* __restrict__ temps,
                                                            const double* __restrict__ nh, int flags, double* __restrict__ out)
{
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= n_lines * n_depth) return;
    const int64_t l = k / n_depth;
    const int d = (int)(k - l * n_depth);
    double g = 0.0;
    if (flags & 8) g = add_rn(g, a_ul[l]);
    if ((flags & 1) && z[l] == 1) {
        const double nu_ = n_effective(ion[l], e_ion[l], e_up[l]);
        const double nl_ = n_effective(ion[l], e_ion[l], e_lo[l]);
        g = add_rn(g, gamma_linear_stark(nu_, nl_, ne[d]));
    }
    if (flags & 2) g = add_rn(g, vald_stark(ne[d], stark[l], temps[d]));
    if (flags & 4) g = add_rn(g, vald_vdw(waals[l], temps[d], mass[l], e_up[l], e_lo[l], nh[d], ion[l], e_ion[l]));
    out[k] = (flags & 16) ? g : g / 2;  // broadening.py:1084; the molecular branch (:771-799) does not halve
}

// ------------------------------------------------------------------------------------------------
// continuum sources (opacities_solvers/base.py:40-317).  Each returns the value for one (depth, nu).
__device__ __forceinline__ double interp1(double x, int n, const double* __restrict__ xp, const double* __restrict__ fp)
{  // np.interp, xp ascending
    if (x <= xp[0]) return fp[0];
    if (x >= xp[n - 1]) return fp[n - 1];
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (xp[mid] <= x) lo = mid; else hi = mid;
    }
    if (xp[lo] == x) return fp[lo];
    const double slope = sub_rn(fp[lo + 1], fp[lo]) / sub_rn(xp[lo + 1], xp[lo]);
    return add_rn(mul_rn(slope, sub_rn(x, xp[lo])), fp[lo]);
}
__device__ __forceinline__ double inv_nu3(double nu) { return 1.0 / mul_rn(mul_rn(nu, nu), nu); }  // nu ** -3

// bf (:178-271): per (level, depth) coefficient BF_CONSTANT * Z**4 * n / n5 (:267-268), precomputed once
__global__ __launch_bounds__(kBlock) void k_bf_coef(int n_depth, int n_species, const int* __restrict__ offs,
                                                    const int* __restrict__ ions, const double* __restrict__ cutoff,
                                                    const double* __restrict__ level_density, double* __restrict__ coef)
{
    const int n_levels = offs[n_species];
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= n_levels * n_depth) return;
    const int L = k / n_depth;
    int s = 0;
    while (s + 1 < n_species && L >= offs[s + 1]) ++s;
    const int zi = ions[s] + 1;
    const double zeff = (double)zi;
    const double r = mul_rn(zeff, sqrt(kRydFreq / cutoff[L]));
    const double r2 = mul_rn(r, r);
    const double n5 = mul_rn(mul_rn(r2, r2), r);  // (...) ** 5
    coef[k] = mul_rn(mul_rn(kBfConst, (double)(zi * zi * zi * zi)), level_density[k]) / n5;
}

struct ContinuumArgs {
    // mirrors sdx_continuum (device pointers)
    const double* lambdas;
    int n_table;
    const double* table_wavelength;
    const double* table_sigma;
    const double* table_density;
    int bf_n_species;
    const int* bf_species_offsets;
    const int* bf_species_ion_number;
    const double* bf_cutoff;
    int bf_n_levels;                 // = bf_species_offsets[bf_n_species] when the host knows it, else 0
    const double* bf_coef;           // [n_levels][n_depth] from k_bf_coef (stand-alone bf source)
    const double* bf_level_density;  // [n_levels][n_depth] (fused total: coefficients formed in LDS)
    int ff_n_species;
    const int* ff_species_ion_number;
    const double* ff_number_density;
    const double* ray_n_h;
    const double* ray_n_he;
    const double* ray_n_h2;
    int rayleigh_enabled;
    const double* electron_density;
    const double* temperature;
    int n_file_planes;             // further tabulated sources as finished planes [n_depth][file_plane_ld], global columns
    const double* file_plane[4];
    int64_t file_plane_ld;
};

__device__ inline double alpha_bf_point(int n_depth, int d, double nu, int n_species, const int* __restrict__ offs,
                                        const double* __restrict__ cutoff, const double* __restrict__ coef)
{
    double total = 0.0;
    for (int s = 0; s < n_species; ++s) {
        double spec = 0.0;  // alpha_spec (:214), levels in plasma order (:221-233)
        for (int L = offs[s]; L < offs[s + 1]; ++L)
            spec = add_rn(spec, nu >= cutoff[L] ? coef[(size_t)L * n_depth + d] : 0.0);
        total = add_rn(total, spec);  // alpha_bf += alpha_spec (:235)
    }
    return mul_rn(total, inv_nu3(nu));  // :237
}
__device__ inline double alpha_ff_point(int n_depth, int d, double nu, double temp, int n_species,
                                        const int* __restrict__ ions, const double* __restrict__ number_density)
{
    double total = 0.0;
    for (int s = 0; s < n_species; ++s) {
        double v = number_density[(size_t)s * n_depth + d] / sqrt(temp);
        v = mul_rn(v, mul_rn(kFfConst, (double)(ions[s] * ions[s])));
        total = add_rn(total, v);
    }
    return mul_rn(total, inv_nu3(nu));
}
__device__ inline double alpha_rayleigh_point(int d, double nu, const double* __restrict__ nh, const double* __restrict__ nhe,
                                              const double* __restrict__ nh2)
{
    double c4 = 0, c6 = 0, c8 = 0;
    if (nh) { c4 = add_rn(c4, mul_rn(20.24, nh[d])); c6 = add_rn(c6, mul_rn(239.2, nh[d])); c8 = add_rn(c8, mul_rn(2256.0, nh[d])); }
    if (nhe) { c4 = add_rn(c4, mul_rn(1.913, nhe[d])); c6 = add_rn(c6, mul_rn(4.52, nhe[d])); c8 = add_rn(c8, mul_rn(7.90, nhe[d])); }
    if (nh2) { c4 = add_rn(c4, mul_rn(28.39, nh2[d])); c6 = add_rn(c6, mul_rn(215.0, nh2[d])); c8 = add_rn(c8, mul_rn(1303.0, nh2[d])); }
    const double nuc = nu > 2.3e15 ? 0.0 : nu;
    const double r = nuc / mul_rn(2.0, mul_rn(kC, kRydCm));
    const double r2 = mul_rn(r, r), r4 = mul_rn(r2, r2);
    const double r6 = mul_rn(r4, r2), r8 = mul_rn(r4, r4);
    return mul_rn(add_rn(add_rn(mul_rn(c4, r4), mul_rn(c6, r6)), mul_rn(c8, r8)), kSigmaT);
}

enum ContSource { kSrcFile1d = 0, kSrcBf = 1, kSrcFf = 2, kSrcRayleigh = 3, kSrcElectron = 4 };

// one source -> out (drop-in calc_alpha_* functions)
__global__ __launch_bounds__(kBlock) void k_continuum_source(int src, int n_depth, int64_t n_nu,
                                                             const double* __restrict__ nus, ContinuumArgs a,
                                                             double* __restrict__ out, int64_t ld)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int d = blockIdx.y;
    if (i >= n_nu) return;
    double v = 0.0;
    switch (src) {
        case kSrcFile1d: v = mul_rn(interp1(a.lambdas[i], a.n_table, a.table_wavelength, a.table_sigma), a.table_density[d]); break;
        case kSrcBf: v = alpha_bf_point(n_depth, d, nus[i], a.bf_n_species, a.bf_species_offsets, a.bf_cutoff, a.bf_coef); break;
        case kSrcFf: v = alpha_ff_point(n_depth, d, nus[i], a.temperature[d], a.ff_n_species, a.ff_species_ion_number, a.ff_number_density); break;
        case kSrcRayleigh: v = alpha_rayleigh_point(d, nus[i], a.ray_n_h, a.ray_n_he, a.ray_n_h2); break;
        case kSrcElectron: v = mul_rn(kSigmaT, a.electron_density[d]); break;
    }
    out[(size_t)d * ld + i] = v;
}

// sigma_file for the 2-D cross-section tables (opacities_solvers/util.py:35-91): scipy's LinearNDInterpolator on the
// table's Delaunay triangulation, evaluated at the mesh (lambda_k, second_d) — second is T (H2+ bf) or 5040/T (H- ff).
// scipy (interpnd.pyx _evaluate_double, qhull.pyx _barycentric_coordinates): barycentric coordinates from Qhull's
// transform of the simplex, c_i = sum_j T[i][j] (x_j - T[2][j]), c_2 = 1 - c_0 - c_1; a point is inside when every
// c >= -eps (eps = 100 DBL_EPSILON); value = sum_j c_j v_j; outside the hull 0 (the reference's fill value).  The
// triangles of a rectilinear table lie two per cell, so the search is a cell look-up and at most two tests.
// scale_kind 1: x 1e-18 (:58); 2: x 1e-26 x k_B x T (:83-88).  zero_rows[d] is raised when a row holds an exact zero
// (the reference's "outside of interpolation range" warning, :59-62, :89-92).
constexpr int kTableAxisMax = 512;
__device__ __forceinline__ int last_le_clipped(const double* a, int n, double v)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return max(0, min(lo - 1, n - 2));
}
__global__ __launch_bounds__(kBlock) void k_sigma_table_2d(int n_x, const double* __restrict__ x_axis, int n_y,
                                                           const double* __restrict__ y_axis, const int* __restrict__ cell_simplices,
                                                           const double* __restrict__ transform,
                                                           const double* __restrict__ simplex_values, int n_depth, int64_t n_nu,
                                                           const double* __restrict__ lambdas, const double* __restrict__ second,
                                                           int scale_kind, const double* __restrict__ temperature,
                                                           double* __restrict__ sigma, int64_t ld, int* __restrict__ zero_rows)
{
    __shared__ double s_x[kTableAxisMax];
    __shared__ int s_zero;
    for (int i = threadIdx.x; i < n_x; i += kBlock) s_x[i] = x_axis[i];
    if (threadIdx.x == 0) s_zero = 0;
    __syncthreads();
    const int d = blockIdx.y;
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k < n_nu) {
        const double qx = lambdas[k], qy = second[d];
        const int i = last_le_clipped(s_x, n_x, qx);
        const int j = last_le_clipped(y_axis, n_y, qy);
        constexpr double eps = 100.0 * 2.220446049250313e-16;
        double val = 0.0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int sidx = cell_simplices[(i * (n_y - 1) + j) * 2 + c];
            const double* t = transform + (size_t)sidx * 6;
            const double dx = sub_rn(qx, t[4]), dy = sub_rn(qy, t[5]);
            const double c0 = add_rn(mul_rn(t[0], dx), mul_rn(t[1], dy));
            const double c1 = add_rn(mul_rn(t[2], dx), mul_rn(t[3], dy));
            const double c2 = sub_rn(sub_rn(1.0, c0), c1);
            if (c0 >= -eps && c1 >= -eps && c2 >= -eps && c0 <= 1.0 + eps && c1 <= 1.0 + eps && c2 <= 1.0 + eps) {
                const double* v = simplex_values + (size_t)sidx * 3;
                val = add_rn(add_rn(mul_rn(c0, v[0]), mul_rn(c1, v[1])), mul_rn(c2, v[2]));
                break;
            }
        }
        if (scale_kind == 1) val = mul_rn(val, 1e-18);
        else if (scale_kind == 2) val = mul_rn(mul_rn(mul_rn(val, 1e-26), kKB), temperature[d]);
        sigma[(size_t)d * ld + k] = val;
        if (val == 0.0) s_zero = 1;
    }
    __syncthreads();
    if (zero_rows && threadIdx.x == 0 && s_zero) zero_rows[d] = 1;
}

__global__ __launch_bounds__(kBlock) void k_rayleigh_clip(int64_t n_nu, double* __restrict__ nus)
{  // base.py:99: tracing_nus[tracing_nus > 2.3e15] = 0, in the caller's array
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n_nu && nus[i] > 2.3e15) nus[i] = 0.0;
}

__global__ __launch_bounds__(kBlock) void k_scale_rows(int n_depth, int64_t n_nu, const double* __restrict__ src, int64_t src_ld,
                                                       const double* __restrict__ density, double* __restrict__ out, int64_t ld)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int d = blockIdx.y;
    if (i < n_nu) out[(size_t)d * ld + i] = mul_rn(src[(size_t)d * src_ld + i], density[d]);
}

__global__ __launch_bounds__(kBlock) void k_scale(int n_depth, int64_t n_nu, double* __restrict__ a, int64_t ld, double factor)
{  // F_nu *= (r[-1] / reference_r)**2, radiation_field_solvers/base.py:340-344
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int d = blockIdx.y;
    if (i < n_nu) a[(size_t)d * ld + i] = mul_rn(a[(size_t)d * ld + i], factor);
}

__global__ __launch_bounds__(kBlock) void k_accumulate(int n_depth, int64_t n_nu, double* __restrict__ total, int64_t tld,
                                                       const double* __restrict__ src, int64_t sld)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int d = blockIdx.y;
    if (i < n_nu) total[(size_t)d * tld + i] = add_rn(total[(size_t)d * tld + i], src[(size_t)d * sld + i]);
}

// fused: total = ((((0 + file) + bf) + ff) + rayleigh) + electron) + line  (calc_alphas order, :655-738,
// then Opacities.calc_total_alphas insertion order, opacities/base.py:24-28).  `line` holds n_split partial
// planes [n_split][n_depth][line_ld] summed here in subset order; line_out (optional) receives that sum.
// The bound-free per-level coefficients of this block's depth are formed in LDS first (k_bf_coef's formula).
template <int NP = 1>
__device__ __forceinline__ void total_alphas_block(const int bx, const int d, int n_depth, int64_t nu_begin, int64_t nu_count,
                                                         const double* __restrict__ nus, ContinuumArgs a,
                                                         const double* __restrict__ line, int64_t line_ld, int n_split,
                                                         double* __restrict__ line_out, int64_t line_out_ld,
                                                         double* __restrict__ total, int64_t total_ld, const bool stage_table = false)
{
    extern __shared__ double s_coef[];  // [n_levels] for depth d, then (stage_table) the 1-D cross-section table
    const int n_levels = a.bf_n_species > 0 ? (a.bf_n_levels > 0 ? a.bf_n_levels : a.bf_species_offsets[a.bf_n_species]) : 0;  // host value: no load to wait for
    // the tabulated cross-section is searched per point: from LDS the bisection costs ~10x less latency than from L2
    double* s_xp = s_coef + n_levels;
    double* s_fp = s_xp + a.n_table;
    if (stage_table && a.table_sigma)
        for (int k = threadIdx.x; k < a.n_table; k += blockDim.x) {
            s_xp[k] = a.table_wavelength[k];
            s_fp[k] = a.table_sigma[k];
        }
    for (int L = threadIdx.x; L < n_levels; L += blockDim.x) {
        int sp = 0;
        while (sp + 1 < a.bf_n_species && L >= a.bf_species_offsets[sp + 1]) ++sp;
        const int zi = a.bf_species_ion_number[sp] + 1;
        const double r = mul_rn((double)zi, sqrt(kRydFreq / a.bf_cutoff[L]));
        const double r2 = mul_rn(r, r);
        const double n5 = mul_rn(mul_rn(r2, r2), r);
        s_coef[L] = mul_rn(mul_rn(kBfConst, (double)(zi * zi * zi * zi)), a.bf_level_density[(size_t)L * n_depth + d]) / n5;
    }
    __syncthreads();
#pragma unroll
    for (int pt = 0; pt < NP; ++pt) {
    const int64_t j = ((int64_t)bx * NP + pt) * blockDim.x + threadIdx.x;
    if (j < nu_count) {
    const int64_t i = nu_begin + j;
    const double nu = nus[i];
    double t = 0.0;
    if (a.table_sigma)
        t = add_rn(t, mul_rn(stage_table ? interp1(a.lambdas[i], a.n_table, s_xp, s_fp) : interp1(a.lambdas[i], a.n_table, a.table_wavelength, a.table_sigma),
                             a.table_density[d]));
    for (int k = 0; k < a.n_file_planes; ++k) t = add_rn(t, a.file_plane[k][(size_t)d * a.file_plane_ld + i]);  // (wave-uniform count and pointers)
    {
        double bf = 0.0;
        for (int sp = 0; sp < a.bf_n_species; ++sp) {
            double spec = 0.0;
            for (int L = a.bf_species_offsets[sp]; L < a.bf_species_offsets[sp + 1]; ++L)
                spec = add_rn(spec, nu >= a.bf_cutoff[L] ? s_coef[L] : 0.0);
            bf = add_rn(bf, spec);
        }
        t = add_rn(t, a.bf_n_species > 0 ? mul_rn(bf, inv_nu3(nu)) : 0.0);
    }
    t = add_rn(t, a.ff_n_species > 0 ? alpha_ff_point(n_depth, d, nu, a.temperature[d], a.ff_n_species, a.ff_species_ion_number, a.ff_number_density) : 0.0);
    if (a.rayleigh_enabled) t = add_rn(t, alpha_rayleigh_point(d, nu, a.ray_n_h, a.ray_n_he, a.ray_n_h2));
    if (a.electron_density) t = add_rn(t, mul_rn(kSigmaT, a.electron_density[d]));
    if (line) {
        double v = line[(size_t)d * line_ld + j];
        for (int s = 1; s < n_split; ++s) v = add_rn(v, line[((size_t)s * n_depth + d) * line_ld + j]);
        if (line_out) line_out[(size_t)d * line_out_ld + j] = v;
        t = add_rn(t, v);
    }
    total[(size_t)d * total_ld + j] = t;
    }
    }
    __syncthreads();  // s_coef may be refilled for another depth by the caller
}

__global__ __launch_bounds__(kBlock) void k_total_alphas(int n_depth, int64_t nu_begin, int64_t nu_count,
                                                         const double* __restrict__ nus, ContinuumArgs a,
                                                         const double* __restrict__ line, int64_t line_ld, int n_split,
                                                         double* __restrict__ line_out, int64_t line_out_ld,
                                                         double* __restrict__ total, int64_t total_ld)
{
    total_alphas_block(blockIdx.x, blockIdx.y, n_depth, nu_begin, nu_count, nus, a, line, line_ld, n_split, line_out, line_out_ld, total,
                       total_ld);
}

// The continuum plane of the fused step, one block per (tile of blockDim frequencies, group of kContDepths depths).  Everything
// that depends on the frequency alone — the tabulated cross-section (a bisection), nu^-3 (a division), the Rayleigh powers,
// which bound-free edges lie below nu — is formed ONCE per frequency and everything that depends on the depth alone — the
// bound-free coefficients (a division and a square root per level), the free-free sum (a division and a square root per
// species), the Rayleigh and Thomson factors — once per block in LDS; a (depth, frequency) point is then ~25 additions and
// multiplications, in calc_alphas' order and rounding (:655-700), where evaluating every point from scratch
// (total_alphas_block) costs ~400 instructions.
constexpr int kContDepths = 8;  // at most; the host picks 1..8 depths per block so that small grids still fill the chip
__device__ __forceinline__ void continuum_tile_block(const int tile, const int dg, const int dgs, int n_depth, int64_t nu_begin, int64_t nu_count,
                                                     const double* __restrict__ nus, ContinuumArgs a, double* __restrict__ cont,
                                                     int64_t cont_ld, const bool stage_table)
{
    extern __shared__ double s_mem[];
    const int n_levels = a.bf_n_species > 0 ? a.bf_n_levels : 0;
    double* s_coef = s_mem;                              // [kContDepths][n_levels]
    double* s_dep = s_coef + kContDepths * n_levels;     // [kContDepths][6]: file density, ff sum, Rayleigh c4 c6 c8, Thomson
    double* s_xp = s_dep + kContDepths * 6;
    double* s_fp = s_xp + a.n_table;
    const int d0 = dg * dgs;
    const int nd = min(dgs, n_depth - d0);
    if (stage_table && a.table_sigma)
        for (int k = threadIdx.x; k < a.n_table; k += blockDim.x) {
            s_xp[k] = a.table_wavelength[k];
            s_fp[k] = a.table_sigma[k];
        }
    for (int k = threadIdx.x; k < nd * n_levels; k += blockDim.x) {
        const int dd = k / n_levels, L = k - dd * n_levels;
        int sp = 0;
        while (sp + 1 < a.bf_n_species && L >= a.bf_species_offsets[sp + 1]) ++sp;
        const int zi = a.bf_species_ion_number[sp] + 1;
        const double r = mul_rn((double)zi, sqrt(kRydFreq / a.bf_cutoff[L]));
        const double r2 = mul_rn(r, r);
        const double n5 = mul_rn(mul_rn(r2, r2), r);
        s_coef[dd * n_levels + L] = mul_rn(mul_rn(kBfConst, (double)(zi * zi * zi * zi)), a.bf_level_density[(size_t)L * n_depth + d0 + dd]) / n5;
    }
    if (threadIdx.x < nd) {
        const int d = d0 + threadIdx.x;
        double* dep = s_dep + threadIdx.x * 6;
        dep[0] = a.table_sigma ? a.table_density[d] : 0.0;
        double ff = 0.0;  // alpha_ff_point's sum (:274-317), the part in front of nu^-3
        for (int sp = 0; sp < a.ff_n_species; ++sp) {
            double v = a.ff_number_density[(size_t)sp * n_depth + d] / sqrt(a.temperature[d]);
            v = mul_rn(v, mul_rn(kFfConst, (double)(a.ff_species_ion_number[sp] * a.ff_species_ion_number[sp])));
            ff = add_rn(ff, v);
        }
        dep[1] = ff;
        double c4 = 0, c6 = 0, c8 = 0;  // alpha_rayleigh_point's per-depth sums (:74-135)
        if (a.rayleigh_enabled) {
            if (a.ray_n_h) { c4 = add_rn(c4, mul_rn(20.24, a.ray_n_h[d])); c6 = add_rn(c6, mul_rn(239.2, a.ray_n_h[d])); c8 = add_rn(c8, mul_rn(2256.0, a.ray_n_h[d])); }
            if (a.ray_n_he) { c4 = add_rn(c4, mul_rn(1.913, a.ray_n_he[d])); c6 = add_rn(c6, mul_rn(4.52, a.ray_n_he[d])); c8 = add_rn(c8, mul_rn(7.90, a.ray_n_he[d])); }
            if (a.ray_n_h2) { c4 = add_rn(c4, mul_rn(28.39, a.ray_n_h2[d])); c6 = add_rn(c6, mul_rn(215.0, a.ray_n_h2[d])); c8 = add_rn(c8, mul_rn(1303.0, a.ray_n_h2[d])); }
        }
        dep[2] = c4, dep[3] = c6, dep[4] = c8;
        dep[5] = a.electron_density ? mul_rn(kSigmaT, a.electron_density[d]) : 0.0;
    }
    __syncthreads();
    const int64_t j = (int64_t)tile * blockDim.x + threadIdx.x;
    if (j >= nu_count) return;
    const int64_t i = nu_begin + j;
    const double nu = nus[i];
    const double sig = a.table_sigma ? (stage_table ? interp1(a.lambdas[i], a.n_table, s_xp, s_fp) : interp1(a.lambdas[i], a.n_table, a.table_wavelength, a.table_sigma)) : 0.0;
    const double inv3 = (a.bf_n_species > 0 || a.ff_n_species > 0) ? inv_nu3(nu) : 0.0;
    double r4 = 0, r6 = 0, r8 = 0;
    if (a.rayleigh_enabled) {
        const double nuc = nu > 2.3e15 ? 0.0 : nu;
        const double r = nuc / mul_rn(2.0, mul_rn(kC, kRydCm));
        const double r2 = mul_rn(r, r);
        r4 = mul_rn(r2, r2), r6 = mul_rn(r4, r2), r8 = mul_rn(r4, r4);
    }
    for (int dd = 0; dd < nd; ++dd) {
        const double* dep = s_dep + dd * 6;
        const double* coef = s_coef + dd * n_levels;
        double t = 0.0;
        if (a.table_sigma) t = add_rn(t, mul_rn(sig, dep[0]));
        for (int k = 0; k < a.n_file_planes; ++k) t = add_rn(t, a.file_plane[k][(size_t)(d0 + dd) * a.file_plane_ld + i]);
        double bf = 0.0;
        for (int sp = 0; sp < a.bf_n_species; ++sp) {
            double spec = 0.0;  // alpha_spec (:214), levels in plasma order (:221-233)
            for (int L = a.bf_species_offsets[sp]; L < a.bf_species_offsets[sp + 1]; ++L) spec = add_rn(spec, nu >= a.bf_cutoff[L] ? coef[L] : 0.0);
            bf = add_rn(bf, spec);
        }
        t = add_rn(t, a.bf_n_species > 0 ? mul_rn(bf, inv3) : 0.0);
        t = add_rn(t, a.ff_n_species > 0 ? mul_rn(dep[1], inv3) : 0.0);
        if (a.rayleigh_enabled) t = add_rn(t, mul_rn(add_rn(add_rn(mul_rn(dep[2], r4), mul_rn(dep[3], r6)), mul_rn(dep[4], r8)), kSigmaT));
        if (a.electron_density) t = add_rn(t, dep[5]);
        cont[(size_t)(d0 + dd) * cont_ld + j] = t;
    }
}

// Frequencies per thread of a continuum block in the fused pre-pass launch (2 was measured: no gain).
// The pre-pass kernels cap their SGPRs at 80: a 1024-thread block is 4 waves per SIMD, and at the 94 SGPRs the compiler
// would use only ONE such block fits a CU (7 waves per SIMD), which ran this launch in three block generations at S-c2
// size (scripts/occupancy_probe.hip: two fit below 64 VGPRs and 81 SGPRs).
constexpr int kContPoints = 1;
// Pre-pass and continuum in ONE launch: the pre-pass is a few latency-bound blocks (binary searches, a grid scan);
// the continuum plane depends on nothing and fills the rest of the chip meanwhile.
template <bool GEN, int LINES>
__global__ __launch_bounds__(kPreBlock) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8))) void k_prepass_continuum(int n_pre_x, int n_pre_y, int cont_tiles, int n_depth, int64_t n_nu,
                                                              const double* __restrict__ nus,
                                                              const double* __restrict__ dnu_partial, int n_partial,
                                                              int64_t n_lines, const double* __restrict__ line_nus,
                                                              const double* __restrict__ doppler,
                                                              const double* __restrict__ gammas, int gamma_cols,
                                                              const double* __restrict__ alphas, LineWork w, int n_line_blocks,
                                                              int64_t nu_begin, int64_t nu_count, ContinuumArgs ca,
                                                              double* __restrict__ cont_plane, int64_t cont_ld, LineParams lp, int stage_table)
{
    const int b = blockIdx.x;
    const int n_pre = n_pre_x * n_pre_y;
    if (b < n_pre) {
        prepass_block<GEN, LINES>(b % n_pre_x, b / n_pre_x, n_pre_y, n_depth, n_nu, nus, dnu_partial, n_partial, n_lines, line_nus, doppler,
                           gammas, gamma_cols, alphas, w, nullptr, nullptr, n_line_blocks, lp);
    } else {
        const int c = b - n_pre;
#ifdef SDX_PRE_STATS
        const unsigned long long st0 = wall_clock64();
#endif
        if (stage_table & 2)  // bit 1: depth-group blocks (the per-depth factors of a group fit LDS); bits 4..7: depths per block
            continuum_tile_block(c % cont_tiles, c / cont_tiles, (stage_table >> 4) & 15, n_depth, nu_begin, nu_count, nus, ca, cont_plane, cont_ld,
                                 (stage_table & 1) != 0);
        else
            total_alphas_block<kContPoints>(c % cont_tiles, c / cont_tiles, n_depth, nu_begin, nu_count, nus, ca, nullptr, 0, 1, nullptr, 0,
                                            cont_plane, cont_ld, (stage_table & 1) != 0);
#ifdef SDX_PRE_STATS
        if (threadIdx.x == 0) {
            unsigned long long* const o = g_pre_stats + (size_t)((kPreStatSlots / 2 + c) & (kPreStatSlots - 1)) * 8;
            o[0] = ((unsigned long long)c << 8) | 4 | 1;
            o[1] = st0;
            o[7] = wall_clock64();
        }
#endif
    }
}

// Culled shards of the fused step: the classification stream (HBM-bound, vector units idle) and the continuum plane (arithmetic,
// no traffic to speak of) in ONE launch of 256-thread blocks — blocks [0, n_dnu) the grid-spacing partials, then the continuum
// tiles, then the n_cls streaming blocks.
__global__ __launch_bounds__(kBlock) void k_classify_continuum(int n_dnu, int n_cls, int n_depth, int64_t n_nu, int64_t n_lines, double* __restrict__ dnu_partial,
                                                               const double* __restrict__ doppler, const double* __restrict__ gammas,
                                                               int gamma_cols, const double* __restrict__ alphas, double* __restrict__ m_max,
                                                               const double* __restrict__ nus, int cont_tiles, int64_t nu_begin, int64_t nu_count,
                                                               ContinuumArgs ca, double* __restrict__ cont_plane, int64_t cont_ld, int stage_table,
                                                               const double* __restrict__ line_nus, int64_t shard_begin, int64_t shard_count,
                                                               int* __restrict__ sel, int64_t cls_begin, int64_t cls_end, FarReq far)
{
    // order of the roles in the grid = order of dispatch: the continuum tiles — chains of dependent work (coefficients, a barrier,
    // a table search), eight depths per block so that they are few and long — go first and run behind the stream (round 4, an
    // eighth of S-c3: this launch 62 us with one depth per block, 46 with eight; tiles first or last: 46 / 47)
    const int n_cont = (int)gridDim.x - n_dnu - n_cls;
    const int b = blockIdx.x;
    if (b < n_dnu) {
        __shared__ double s_red[kBlock / 64];
        dnu_partial_block(b, n_dnu, n_nu, nus, dnu_partial, s_red, far);
    } else if (b < n_dnu + n_cont) {
        const int c = b - n_dnu;
        continuum_tile_block(c % cont_tiles, c / cont_tiles, (stage_table >> 4) & 15, n_depth, nu_begin, nu_count, nus, ca, cont_plane, cont_ld,
                             (stage_table & 1) != 0);
    } else {
        const int k = b - n_dnu - n_cont;
        if (sel && k == n_cls - 1) shard_range(n_nu, nus, n_lines, line_nus, shard_begin, shard_count, sel);  // (four threads, on the side)
        classify_block(k, n_cls, n_depth, cls_begin, cls_end, doppler, gammas, gamma_cols, alphas, m_max);
    }
}

// ------------------------------------------------------------------------------------------------
// Formal solution (radiation_field_solvers/base.py:85-346).  A group of G adjacent lanes of one wave
// owns one frequency (64/G groups per wave); lane g of the group traces angles g, g+G, ... (at most P of
// them) through all depth gaps, keeping the rolling state — two mean opacities, three source values, one
// intensity per angle — in registers.  After every gap the group sums I_theta * w_theta in ascending-theta
// order with wave shuffles and lane 0 adds it to F_nu[gap+1] (:336-338).  Geometric-mean opacity (:121),
// tau (:123-129), Planck source (:133) and the weights (:138) are formed with the reference's operations;
// the mean opacity and source are shared by the angles a lane owns instead of being recomputed per angle.
template <int P>
__global__ __launch_bounds__(kBlock) void k_raytrace_basic(int n_depth, int64_t n_nu, int n_theta, int theta_stride, int G,
                                                     const double* __restrict__ nus, const double* __restrict__ temps,
                                                     const double* __restrict__ ray_dist, const double* __restrict__ wts,
                                                     const double* __restrict__ alphas, int64_t ald, double* __restrict__ F,
                                                     int64_t fld, double* __restrict__ I_nus, int accumulate)
{
    // n_theta angles are traced here; ray_dist / I_nus rows have theta_stride entries (a chunk of a longer list)
    const int lane = threadIdx.x & 63;
    const int gpw = 64 / G;  // groups per wave
    const int grp = lane / G, g = lane - grp * G;
    const int64_t i = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * gpw + grp;
    const bool valid = grp < gpw && i < n_nu;
    const int64_t ic = i < n_nu ? i : n_nu - 1;
    const int lane0 = lane - g;
    const int n_gap = n_depth - 1;
    const double nu = nus[ic];

    double la0 = log(alphas[ic]);                                 // log alpha[gap]
    double la1 = log(alphas[(size_t)ald + ic]);                   // log alpha[gap+1]
    double mean0 = exp(mul_rn(add_rn(la1, la0), 0.5));            // :121
    double s0 = planck(nu, temps[0]), s1 = planck(nu, temps[1]);  // :133
    double inten[P], wt[P];
    int th[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        inten[k] = 0.0;  // I[0] = 0 (:134-136)
        th[k] = g + k * G;
        wt[k] = th[k] < n_theta ? wts[th[k]] : 0.0;
        if (valid && I_nus && th[k] < n_theta) I_nus[(size_t)i * theta_stride + th[k]] = 0.0;
    }

    for (int gap = 0; gap < n_gap; ++gap) {
        const bool last = gap == n_gap - 1;
        double mean1 = 0.0, s2 = 0.0, la2 = 0.0;
        if (!last) {
            la2 = log(alphas[(size_t)(gap + 2) * ald + ic]);
            mean1 = exp(mul_rn(add_rn(la2, la1), 0.5));
            s2 = planck(nu, temps[gap + 2]);
        }
        double fsum = 0.0;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            if (th[k] < n_theta) {
                const double tau0 = mul_rn(mean0, ray_dist[(size_t)gap * theta_stride + th[k]]);
                double inew;
                if (tau0 == 0.0) {
                    inew = inten[k];  // :203-206, :253-254
                } else {
                    double w0, w1, w2;
                    rt_weights(tau0, w0, w1, w2);
                    if (!last) {  // :208-249
                        const double tau1 = mul_rn(mean1, ray_dist[(size_t)(gap + 1) * theta_stride + th[k]]);
                        const double sum01 = add_rn(tau0, tau1);
                        const double second =
                            mul_rn(w1, sub_rn(mul_rn(sub_rn(s1, s2), tau0 / tau1), mul_rn(sub_rn(s1, s0), tau1 / tau0))) / sum01;
                        const double third = mul_rn(w2, add_rn(sub_rn(s2, s1) / tau1, sub_rn(s0, s1) / tau0)) / sum01;
                        inew = add_rn(add_rn(add_rn(mul_rn(sub_rn(1.0, w0), inten[k]), mul_rn(w0, s1)), second), third);
                    } else {  // :256-266
                        const double third = mul_rn(w2, sub_rn(s0, s1)) / mul_rn(tau0, tau0);
                        inew = add_rn(add_rn(mul_rn(sub_rn(1.0, w0), inten[k]), mul_rn(w0, s1)), third);
                    }
                }
                inten[k] = inew;
                if (valid && I_nus) I_nus[((size_t)(gap + 1) * n_nu + i) * theta_stride + th[k]] = inew;
            }
        }
        if (F) {
            if (G == 1) {
#pragma unroll
                for (int k = 0; k < P; ++k)
                    if (th[k] < n_theta) fsum = add_rn(fsum, mul_rn(inten[k], wt[k]));
            } else {
                // ascending theta = gg + k*G: k outer, group lanes inner
#pragma unroll
                for (int k = 0; k < P; ++k) {
                    const double mine = mul_rn(inten[k], wt[k]);
                    for (int gg = 0; gg < G; ++gg) {
                        const double v = __shfl(mine, lane0 + gg);
                        if (gg + k * G < n_theta) fsum = add_rn(fsum, v);
                    }
                }
            }
            if (valid && g == 0) {
                double* p = F + (size_t)(gap + 1) * fld + i;
                *p = accumulate ? add_rn(*p, fsum) : fsum;
                if (gap == 0 && !accumulate) F[i] = 0.0;
            }
        }
        la0 = la1; la1 = la2; mean0 = mean1; s0 = s1; s1 = s2;
    }
}

// ------------------------------------------------------------------------------------------------
// Formal solution, LDS-staged (the default).  Same lane <-> (frequency, angle) mapping as k_raytrace_basic:
// a group of G adjacent lanes owns one frequency, lane g traces angle(s) g, g+G, ...  What is shared is
// prepared once and kept in LDS:
//   block    the ray-length table ray_dist[gap][theta] (:302-305) and its reciprocals;
//   group    phase 1: the G lanes split the N_d depth points: log(alpha) and the Planck source S (:133) -> LDS,
//            then per gap the geometric-mean opacity exp((log a[g+1] + log a[g]) * 0.5) (:121) and its reciprocal;
//   lane     phase 2: walks the gaps for its own angle(s): tau = mean * ray_dist (:123-129, the reference's
//            product), weights (:22-45), second-order recurrence (:200-266).  The divisions of :208-242 are
//            re-expressed with the slopes a = (S[g+2]-S[g+1])/tau[g+1], b = (S[g]-S[g+1])/tau[g]:
//                second = w1 (b tau[g+1] - a tau[g]) / (tau[g] + tau[g+1]),   third = w2 (a + b) / (tau[g] + tau[g+1])
//            with 1/tau formed as (1/mean)(1/ray_dist) — shared across angles / frequencies — and one reciprocal
//            per step for the sum.  tau = 0 gives the same inf/NaN pattern as the reference's unguarded divisions.
//   flux     I_theta * w_theta goes to LDS; every kBatch gaps the wave sums each (gap, frequency) over theta in
//            ascending order (the reference's order, :324-338) and writes F_nu.

// Optional fusion of Opacities.calc_total_alphas into the raytrace's column staging: total = continuum + line,
// line = sum of the partial planes in subset order (the same additions k_total_alphas performs).
struct FusedTotal {
    const double* cont;    // [n_depth][cld] continuum in calc_alphas order, or nullptr: read `alphas` instead
    int64_t cld;
    const double* planes;  // [n_planes][n_depth][pld] partial line-opacity planes, or nullptr (no lines)
    int n_planes;
    int64_t pld;
    double* total_out;     // [n_depth][out_ld], optional
    double* line_out;      // optional
    int64_t out_ld;
    // source function given by the caller as a plane [n_depth][sld] (RadiationField.source_function is any callable
    // (nu, T) -> (N_d, N_nu), radiation_field_solvers/base.py:133); nullptr: the Planck function, evaluated here
    const double* source;
    int64_t sld;
    // further line-opacity planes [n_depth][eld] the caller has formed (the molecular list of include_molecules,
    // opacities_solvers/base.py:716-736), added after the step's own line opacity in this order: total = (cont + line) + extra...
    const double* extra[2];
    int n_extra;
    int64_t eld;
};

// Formal solution, LDS-staged (grids that fill the chip; every geometry).  Lane <-> (frequency, angle): a group of G adjacent
// lanes owns one frequency, lane g traces angle(s) g, g + G, ...  Staged once and kept in LDS:
//   block    the ray-length table ray_dist[gap][theta] (:302-305);
//   wave     per frequency of the wave, the G lanes of its group split the N_d depth points: the source function S (:133;
//            planck_staged, or the caller's plane) and sqrt(alpha).  The geometric-mean opacity of a gap (:121,
//            exp((log a[g+1] + log a[g]) / 2)) is the product of the two square roots at its ends — one correctly rounded
//            square root per point instead of a logarithm per point and an exponential per gap, and within 1e-15 of the
//            reference's value (whose own error is ~|log alpha| ulp);
//   lane     walks the gaps for its own angle(s): tau = mean * ray_dist (:123-129), then one rt_coef (sdx_math.h: the
//            step as an affine map of the incoming intensity, one reciprocal, the reference's weights) and one FMA;
//   flux     I_theta * w_theta goes to the wave's LDS; every kBatch gaps the wave sums each (gap, frequency) over theta in
//            two ascending halves and writes F_nu.
// Only the ray table is shared by the block: after the staging barrier the waves never meet again (wave-level hand-overs).
constexpr int kRtBlock = 256;
template <int P>
__global__ __launch_bounds__(kRtBlock) void k_raytrace(int n_depth, int64_t n_nu, int n_theta, int theta_stride, int G,
                                                     const double* __restrict__ nus, const double* __restrict__ temps,
                                                     const double* __restrict__ ray_dist, const double* __restrict__ wts,
                                                     const double* __restrict__ alphas, int64_t ald, double* __restrict__ F,
                                                     int64_t fld, double* __restrict__ I_nus, int accumulate, int inward, int gpw, FusedTotal ft)
{
    constexpr int kBatch = P == 1 ? 4 : 2;
    extern __shared__ double smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // gpw = groups (frequencies) per wave, <= 64 / G; the host lowers it when the LDS columns would not fit
    const int grp = lane / G, g = lane - grp * G;
    const int TH = P * G;  // theta slots per group, ascending theta = k*G + g
    const int64_t i0 = ((int64_t)blockIdx.x * (kRtBlock / 64) + wave) * gpw;  // first frequency of this wave
    const int64_t i = i0 + grp;
    const bool active = grp < gpw;
    const bool valid = active && i < n_nu;
    const int64_t ic = i < n_nu ? i : n_nu - 1;
    const int n_gap = n_depth - 1;
    const int col = n_depth;  // LDS row stride per group
    double* wbase = smem + (size_t)wave * (2 * gpw * col + kBatch * gpw * TH);
    double2* sP = (double2*)wbase;            // (source function, sqrt(alpha)) [gpw][col]: a point's pair is ONE 16-byte LDS read
    double* sX = wbase + 2 * gpw * col;       // flux terms       [kBatch][gpw][TH]
    const double nu = nus[ic];

    // (Staging the columns a batch of G depth points AHEAD of the recurrence — the loads of batch b + 1 requested when batch b is
    // committed, so that a shard's single generation of workgroups does not wait for all its planes before any wave computes — was
    // built and measured in round 5: k_raytrace 60 - 91 us instead of 55 - 84 on the eighths of S-c3 and 428 instead of 377 on the
    // whole grid.  The commit inside the gap loop needs the loop's registers: 46 spilled VGPRs at seven waves per SIMD, and a lone
    // generation is bound by the 55-step chain of each wave, not by its staging.  Removed.)
    if (active) {
        for (int d = g; d < n_depth; d += G) {
            double a;
            if (ft.cont) {
                a = ft.cont[(size_t)d * ft.cld + ic];
                if (ft.planes) {
                    double line = ft.planes[(size_t)d * ft.pld + ic];
                    for (int sp = 1; sp < ft.n_planes; ++sp) line = add_rn(line, ft.planes[((size_t)sp * n_depth + d) * ft.pld + ic]);
                    a = add_rn(a, line);
                    if (valid && ft.line_out) ft.line_out[(size_t)d * ft.out_ld + i] = line;
                }
                for (int x = 0; x < ft.n_extra; ++x) a = add_rn(a, ft.extra[x][(size_t)d * ft.eld + ic]);
                if (valid && ft.total_out) ft.total_out[(size_t)d * ft.out_ld + i] = a;
            } else {
                a = alphas[(size_t)d * ald + ic];
            }
            sP[grp * col + d] = double2{ft.source ? ft.source[(size_t)d * ft.sld + ic] : planck_staged(nu, temps[d]), sqrt(a)};
        }
    }
    wave_sync();  // (nothing is shared between the waves of a block any more: the ray table is read from L1)

    const int gi = (active ? grp : 0) * col;  // idle lanes shadow group 0 and never store
    double inten[P], wt[P];
    int th[P];
    bool on[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        th[k] = min(g + k * G, n_theta - 1);
        on[k] = g + k * G < n_theta;
        inten[k] = 0.0;  // np.zeros (:134)
        wt[k] = on[k] ? wts[th[k]] : 0.0;
    }
    if (inward) {
        // spherical geometry: sweep from the surface to the innermost point first (:141-198).  Only I[0] of this
        // sweep survives (the outward pass overwrites the other rows); gap 0 wraps to the LAST gap / depth exactly
        // as the reference's negative index does.  Either optical depth zero: no change (:146-149).
        for (int gap = n_gap - 1; gap >= 0; --gap) {
            const int gm = gap > 0 ? gap - 1 : n_gap - 1;
            const int dm = gap > 0 ? gap - 1 : n_depth - 1;
            const double2 q0 = sP[gi + gap + 1], q1 = sP[gi + gap], q2 = sP[gi + dm];
            const double s0 = q0.x, s1 = q1.x, s2 = q2.x;
            const double mg = q1.y * q0.y, mm = sP[gi + gm].y * sP[gi + gm + 1].y;
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const double tg = mul_rn(mg, ray_dist[(size_t)gap * theta_stride + th[k]]), tm = mul_rn(mm, ray_dist[(size_t)gm * theta_stride + th[k]]);
                double c, e;
                rt_coef<false>(tg, tm, s0 - s1, s2 - s1, s1, c, e);
                inten[k] = tm == 0.0 ? inten[k] : fma(c, inten[k], e);
            }
        }
#pragma unroll
        for (int k = 0; k < P; ++k) {
            if (valid && I_nus && on[k]) I_nus[(size_t)i * theta_stride + th[k]] = inten[k];
            if (active) sX[grp * TH + k * G + g] = inten[k] * wt[k];
        }
        wave_sync();
        if (F && lane < gpw && i0 + lane < n_nu) {
            double sum = 0.0;
            for (int t = 0; t < n_theta; ++t) sum = add_rn(sum, sX[lane * TH + t]);
            double* dst = F + i0 + lane;
            *dst = accumulate ? add_rn(*dst, sum) : sum;
        }
        wave_sync();
    } else {
#pragma unroll
        for (int k = 0; k < P; ++k)
            if (valid && I_nus && on[k]) I_nus[(size_t)i * theta_stride + th[k]] = 0.0;
        if (valid && g == 0 && F && !accumulate) F[i] = 0.0;
    }
    // rolling state: optical depth of the current gap per angle, source at its two ends, sqrt(alpha) at its far end
    // (the ray-length table — 9 KB at 56 depths x 20 angles — is read from global memory, i.e. the L1: a lane needs ONE entry
    // per gap, requested a gap ahead; staged in LDS it cost every block 9 KB, and with them two of seven workgroups per CU)
    double tau0[P], rd_next[P];
    const double* rdp[P];
    const double2 p0 = sP[gi], p1 = sP[gi + 1];
    double a1 = p1.y;
    {
        const double mean0 = p0.y * a1;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            rdp[k] = ray_dist + th[k];
            tau0[k] = mul_rn(mean0, *rdp[k]);
            rdp[k] += n_gap > 1 ? theta_stride : 0;
            rd_next[k] = *rdp[k];  // gap 1
        }
    }
    double s1 = p1.x, d10 = p0.x - s1;
    const float inv_gpw = 1.0f / (float)gpw;

    for (int gap0 = 0; gap0 < n_gap; gap0 += kBatch) {
        const int nb = min(kBatch, n_gap - gap0);
        for (int b = 0; b < nb; ++b) {
            const int gap = gap0 + b;
            if (gap < n_gap - 1) {  // :208-249
                const double2 p2 = sP[gi + gap + 2];
                const double s2 = p2.x, a2 = p2.y;
                const double mean1 = a1 * a2, d21 = s2 - s1;
#pragma unroll
                for (int k = 0; k < P; ++k) {
                    const double t1 = mul_rn(mean1, rd_next[k]);
                    rdp[k] += gap + 2 < n_gap ? theta_stride : 0;
                    rd_next[k] = *rdp[k];  // gap + 2, for the next trip
                    double c, e;
                    rt_coef<false>(tau0[k], t1, d10, d21, s1, c, e);
                    const double inew = fma(c, inten[k], e);
                    inten[k] = inew;
                    tau0[k] = t1;
                    if (valid && I_nus && on[k]) I_nus[((size_t)(gap + 1) * n_nu + i) * theta_stride + th[k]] = inew;
                    if (active) sX[(b * gpw + grp) * TH + k * G + g] = inew * wt[k];
                }
                d10 = -d21, s1 = s2, a1 = a2;
            } else {  // the final gap (:253-266)
#pragma unroll
                for (int k = 0; k < P; ++k) {
                    double c, e;
                    rt_coef<true>(tau0[k], 0.0, d10, 0.0, s1, c, e);
                    const double inew = fma(c, inten[k], e);
                    inten[k] = inew;
                    if (valid && I_nus && on[k]) I_nus[((size_t)(gap + 1) * n_nu + i) * theta_stride + th[k]] = inew;
                    if (active) sX[(b * gpw + grp) * TH + k * G + g] = inew * wt[k];
                }
            }
        }
        wave_sync();
        if (F) {
            // flux of the nb gaps: lanes <-> (gap, frequency, half of the angles); each half is summed in ascending theta
            // (the reference's order, :324-338) and the lower half is added to the upper one
            const int half = (n_theta + 1) >> 1;
            for (int p = lane; p < 2 * nb * gpw; p += 64) {
                const int h = p & 1, q = p >> 1;
                const int b = (int)(((float)q + 0.5f) * inv_gpw), gq = q - b * gpw;  // q / gpw without an integer division (q < 2^20: exact)
                const double* c = sX + (b * gpw + gq) * TH + (h ? half : 0);
                const int cnt = h ? n_theta - half : half;
                double sum = 0.0;
                int t = 0;
                for (; t + 5 <= cnt; t += 5) {  // five terms a trip (their loads in flight together), added in ascending order
                    const double c0 = c[t], c1 = c[t + 1], c2 = c[t + 2], c3 = c[t + 3], c4 = c[t + 4];
                    sum = add_rn(add_rn(add_rn(add_rn(add_rn(sum, c0), c1), c2), c3), c4);
                }
                for (; t < cnt; ++t) sum = add_rn(sum, c[t]);
                const double other = __shfl_xor(sum, 1);  // 2 nb gpw is even: the partner lane is in the loop too
                const int64_t iq = i0 + gq;
                if (h == 0 && iq < n_nu) {
                    const double tot = add_rn(sum, other);
                    double* dst = F + (size_t)(gap0 + b + 1) * fld + iq;
                    *dst = accumulate ? add_rn(*dst, tot) : tot;
                }
            }
        }
        wave_sync();
    }
}

// The formal solution of the fp32-mixed TOLERANCE path (mixed_precision = 1; plane-parallel, one angle per lane, flux only): the
// layout of k_raytrace<1> — lane <-> (frequency, angle), columns staged per wave — with the recurrence in fp32 (rt_coef32:
// hardware exp2 / rcp, ~27 instructions per step at twice the fp64 issue rate against ~54).  sqrt(alpha), the source function, its
// differences between adjacent points (formed as differences: planck32_pair) and the ray lengths are staged as floats (the totals
// are formed in fp64 and rounded once); F_nu is written as doubles.
// Against the fp64 kernels: < 1e-5 of the flux on the full-size workloads (tests/test_gpu_configs.py, stated tolerance 1e-4).
constexpr int kRt32Batch = 8;  // gaps per flux reduction of k_raytrace_f32 (6 measured: no gain from the eighth block a CU's LDS then holds)
__global__ __launch_bounds__(kRtBlock) void k_raytrace_f32(int n_depth, int64_t n_nu, int n_theta, int theta_stride, int G,
                                                           const double* __restrict__ nus, const double* __restrict__ temps,
                                                           const double* __restrict__ ray_dist, const double* __restrict__ wts,
                                                           const double* __restrict__ alphas, int64_t ald, double* __restrict__ F,
                                                           int64_t fld, int gpw, FusedTotal ft)
{
    constexpr int kBatch = kRt32Batch;
    extern __shared__ double smem[];
    float* fmem = (float*)smem;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = lane / G, g = lane - grp * G;
    const int64_t i0 = ((int64_t)blockIdx.x * (kRtBlock / 64) + wave) * gpw;
    const int64_t i = i0 + grp;
    const bool active = grp < gpw;
    const bool valid = active && i < n_nu;
    const int64_t ic = i < n_nu ? i : n_nu - 1;
    const int n_gap = n_depth - 1, col = n_depth;
    float* sRD = fmem;                                   // ray_dist [n_gap][n_theta], shared by the block
    float* sK = sRD + n_gap * n_theta;                   // 1 / (k T_d) [col], shared by the block
    float* sR = sK + col;                                // (T_d - T_{d+1}) / T_{d+1} [col]: the difference formed in fp64
    // per wave: what a step of the recurrence needs of its NEXT point as ONE 16-byte LDS read — sP[k] = (sqrt(alpha_{k+1}), S_{k+1},
    // S_k - S_{k+1}, -) for k < n_gap, the difference formed as a difference (planck32_pair), not from the two rounded values; the
    // first point's (sqrt(alpha_0), S_0) in the spare slot k = n_gap
    const int shared_floats = (n_gap * n_theta + 2 * col + 3) & ~3;
    float* wbase = fmem + shared_floats + (size_t)wave * (4 * gpw * col + kBatch * gpw * G);
    float4* sP = (float4*)wbase;                         // [gpw][col]
    float* sX = wbase + 4 * gpw * col;                   // flux terms [kBatch][gpw][G]
    const double nu = nus[ic];
    for (int k = threadIdx.x; k < n_gap * n_theta; k += kRtBlock) {
        const int gp = k / n_theta, t = k - gp * n_theta;
        sRD[k] = (float)ray_dist[(size_t)gp * theta_stride + t];
    }
    for (int k = threadIdx.x; k < n_depth; k += kRtBlock) {
        const double t = temps[k], tn = temps[min(k + 1, n_depth - 1)];
        sK[k] = __builtin_amdgcn_rcpf((float)mul_rn(kKB, t));
        sR[k] = (float)sub_rn(t, tn) * __builtin_amdgcn_rcpf((float)tn);
    }
    __syncthreads();
    if (active) {
        const float hn = (float)mul_rn(kH, nu);
        const float pre = (float)(mul_rn(mul_rn(2.0, kH), mul_rn(mul_rn(nu, nu), nu)) * (1.0 / (kC * kC)));
        for (int d = g; d < n_depth; d += G) {
            double a;
            if (ft.cont) {
                a = ft.cont[(size_t)d * ft.cld + ic];
                if (ft.planes) {
                    double line = ft.planes[(size_t)d * ft.pld + ic];
                    for (int sp = 1; sp < ft.n_planes; ++sp) line = add_rn(line, ft.planes[((size_t)sp * n_depth + d) * ft.pld + ic]);
                    a = add_rn(a, line);
                    if (valid && ft.line_out) ft.line_out[(size_t)d * ft.out_ld + i] = line;
                }
                for (int x = 0; x < ft.n_extra; ++x) a = add_rn(a, ft.extra[x][(size_t)d * ft.eld + ic]);
                if (valid && ft.total_out) ft.total_out[(size_t)d * ft.out_ld + i] = a;
            } else {
                a = alphas[(size_t)d * ald + ic];
            }
            const float sa = __builtin_sqrtf((float)a);
            const int dn = min(d + 1, n_depth - 1);
            float sv, dv;
            if (ft.source) {
                const double s0 = ft.source[(size_t)d * ft.sld + ic];
                sv = (float)s0, dv = (float)sub_rn(s0, ft.source[(size_t)dn * ft.sld + ic]);
            } else {
                planck32_pair(hn, pre, sK[d], sK[dn], sR[d], sv, dv);
            }
            float* const mine = (float*)(sP + grp * col + (d > 0 ? d - 1 : n_gap));
            mine[0] = sa, mine[1] = sv;
            ((float*)(sP + grp * col + d))[2] = dv;  // (slot n_gap: the last point's 0, never read)
        }
    }
    __syncthreads();
    const int gi = (active ? grp : 0) * col;
    const int th = min(g, n_theta - 1);
    const float wt = g < n_theta ? (float)wts[th] : 0.f;
    if (valid && g == 0) F[i] = 0.0;
    float inten = 0.f;
    const float4 p0 = sP[gi];
    float a1 = p0.x;
    float t0 = (sP[gi + n_gap].x * a1) * sRD[th];
    float s1 = p0.y, d10 = p0.z;
    const float inv_gpw = 1.0f / (float)gpw;
    for (int gap0 = 0; gap0 < n_gap; gap0 += kBatch) {
        const int nb = min(kBatch, n_gap - gap0);
        for (int b = 0; b < nb; ++b) {
            const int gap = gap0 + b;
            const bool last = gap == n_gap - 1;
            const int nx = last ? gap : gap + 1;
            const float4 p = sP[gi + nx];
            const float s2 = p.y, a2 = p.x;
            const float t1 = (a1 * a2) * sRD[nx * n_theta + th];
            const float d21 = last ? 0.f : -p.z;
            float c, e;
            rt_coef32(t0, t1, d10, d21, s1, last, c, e);
            inten = fmaf(c, inten, e);
            if (active) sX[(b * gpw + grp) * G + g] = inten * wt;
            t0 = t1, d10 = -d21, s1 = s2, a1 = a2;
        }
        wave_sync();
        {
            const int half = (n_theta + 1) >> 1;
            for (int p = lane; p < 2 * nb * gpw; p += 64) {
                const int h = p & 1, q = p >> 1;
                const int b = (int)(((float)q + 0.5f) * inv_gpw), gq = q - b * gpw;
                const float* cc = sX + (b * gpw + gq) * G + (h ? half : 0);
                const int cnt = h ? n_theta - half : half;
                float sum = 0.f;
                int t = 0;
                for (; t + 5 <= cnt; t += 5) {
                    const float c0 = cc[t], c1 = cc[t + 1], c2 = cc[t + 2], c3 = cc[t + 3], c4 = cc[t + 4];
                    sum = ((((sum + c0) + c1) + c2) + c3) + c4;
                }
                for (; t < cnt; ++t) sum += cc[t];
                const float other = __shfl_xor(sum, 1);
                const int64_t iq = i0 + gq;
                if (h == 0 && iq < n_nu) F[(size_t)(gap0 + b + 1) * fld + iq] = (double)(sum + other);
            }
        }
        wave_sync();
    }
}

// Formal solution with the GAPS OF A RAY SPLIT OVER THE WAVES OF A WORKGROUP (plane-parallel geometry, one angle per lane).
// k_raytrace above is bounded by latency, not throughput: a wave walks all N_d - 1 gaps of its rays one after the other —
// ~5000 dependent-issue instructions — and a grid of a few thousand frequencies offers only 2-3 such waves per SIMD.  But a
// step of the recurrence is AFFINE in the incoming intensity, I' = c I + e with c = 1 - w0, and everything expensive
// (exp(-tau), the weights, the second-order source terms: ~55 of ~60 instructions) is in c and e, which do not depend on I.
// So the NS waves of a workgroup take the same rays (lane <-> (frequency, angle), as above) and one SEGMENT of L = ceil
// (n_gap / NS) gaps each:
//   1  every wave forms (c, e) of its gaps — kept in registers — and composes them, (A, B) <- (c A, c B + e);
//   2  the segment maps meet in LDS; wave s folds the maps of the segments below it: its incoming intensity;
//   3  it replays its gaps, I <- c I + e (one FMA each), and writes I w_theta to LDS; the flux of its own gaps is summed
//      over theta by the wave itself, in the order of k_raytrace (two ascending halves).
// Eight times the waves, each an eighth as long.  Staging (ray table and its reciprocals, log alpha, Planck source, geometric means) is done once
// per workgroup by all its threads; the flux terms reuse that LDS after the barrier of step 2.
// The intensity differs from k_raytrace's by the rounding of the composition (a few ulp; the tolerance is 1e-10).
template <int NS, int LMAX>
__global__ __launch_bounds__(64 * NS) __attribute__((amdgpu_waves_per_eu(NS >= 8 ? 6 : 4, 8))) void k_raytrace_seg(
    int n_depth, int64_t n_nu, int n_theta, int theta_stride, const double* __restrict__ nus, const double* __restrict__ temps,
    const double* __restrict__ ray_dist, const double* __restrict__ wts, const double* __restrict__ alphas,
    int64_t ald, double* __restrict__ F, int64_t fld, double* __restrict__ I_nus, int gpw, FusedTotal ft)
{
    extern __shared__ double smem[];
    const int lane = threadIdx.x & 63;
    const int seg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: gap indices and their branches stay scalar
    const int G = n_theta;
    const int grp = lane / G, g = lane - grp * G;
    // XCD-aware order: a workgroup reads and writes only gpw (= 3) adjacent columns of every plane row, less than half a 64-byte
    // sector — the workgroups that share a sector must share an L2, or every XCD fetches and writes back its own copy of it
    // (measured: 30 MB fetched, 13 MB written for 10 + 7 MB).  Workgroups b, b + 8, ... run on one XCD: they take one
    // contiguous eighth of the frequencies (the work per frequency is uniform here).
    const int64_t n_wg = (n_nu + gpw - 1) / gpw, per_xcd = (n_wg + 7) / 8;
    const int64_t wg = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= per_xcd || wg >= n_wg) return;  // (the whole workgroup: no barrier is left waiting)
    const int64_t i0 = wg * gpw;
    const int64_t i = i0 + grp;
    const bool active = grp < gpw;
    const bool valid = active && i < n_nu;
    const int n_gap = n_depth - 1, col = n_depth;
    const int rstride = n_gap | 1;             // odd row stride: the angles of a wave read distinct banks
    double* sAB = smem;                        // [NS][64][2] segment maps
    double* sRT = sAB + NS * 128;              // ray_dist TRANSPOSED [n_theta][rstride]: a lane's gaps are consecutive (immediate offsets)
    // (source function, sqrt(alpha)) [gpw][col], a point's pair ONE 16-byte LDS read; the geometric-mean opacity of a gap (:121) is
    // the product of its two ends' square roots
    double2* sP = (double2*)(sRT + ((n_theta * rstride + 1) & ~1));
    double* sFx = sRT;                         // after the barrier of step 2: flux terms [NS][LMAX][gpw][G]

    // staging without a division per item: lane <-> (one of 64 / n_theta gaps, angle) once; then a wave takes one frequency's
    // column (lane <-> depth) — waves 0 .. gpw-1 sqrt(alpha), the next gpw the source function
    {
        const int per = 64 / G;  // gaps per wave and trip
        if (grp < per)
            for (int gp = seg * per + grp; gp < n_gap; gp += NS * per) sRT[g * rstride + gp] = ray_dist[(size_t)gp * theta_stride + g];
    }
    for (int part = seg; part < 2 * gpw; part += NS) {
        const bool source = part >= gpw;
        const int gq = source ? part - gpw : part;
        const int64_t iq = i0 + gq;
        const bool vq = iq < n_nu;
        const int64_t ic = vq ? iq : n_nu - 1;
        if (source) {
            const double nu = nus[ic];
            for (int d = lane; d < n_depth; d += 64) sP[gq * col + d].x = ft.source ? ft.source[(size_t)d * ft.sld + ic] : planck_staged(nu, temps[d]);
            continue;
        }
        for (int d = lane; d < n_depth; d += 64) {
            double a;
            if (ft.cont) {
                a = ft.cont[(size_t)d * ft.cld + ic];
                if (ft.planes) {
                    double line = ft.planes[(size_t)d * ft.pld + ic];
                    for (int sp = 1; sp < ft.n_planes; ++sp) line = add_rn(line, ft.planes[((size_t)sp * n_depth + d) * ft.pld + ic]);
                    a = add_rn(a, line);
                    if (vq && ft.line_out) ft.line_out[(size_t)d * ft.out_ld + iq] = line;
                }
                for (int x = 0; x < ft.n_extra; ++x) a = add_rn(a, ft.extra[x][(size_t)d * ft.eld + ic]);
                if (vq && ft.total_out) ft.total_out[(size_t)d * ft.out_ld + iq] = a;
            } else {
                a = alphas[(size_t)d * ald + ic];
            }
            sP[gq * col + d].y = sqrt(a);
        }
    }
    __syncthreads();

    // step 1: the coefficients of this wave's gaps [g_lo, g_lo + count).  The segments are aligned to the END of the ray — the first
    // wave takes the short one — and a wave's gaps to the end of its LMAX register slots (slot j <-> gap g_lo + j - j0): the final
    // gap, the one step with a formula of its own (:253-266), is then always slot LMAX - 1 of the last wave, a static place
    const int L = (n_gap + NS - 1) / NS;
    const int g_end = n_gap - (NS - 1 - seg) * L;
    const int g_lo = max(0, g_end - L);
    const int count = max(0, g_end - g_lo);
    const int j0 = LMAX - count;
    const int gi = (active ? grp : 0) * col;  // idle lanes shadow group 0 and never store
    const int th = min(g, n_theta - 1);
    const double wt = wts[th];
    double c[LMAX], e[LMAX];
    double A = 1.0, B = 0.0;
    // the unrolled pass carries no per-lane exception handling: a lane with anything unusual (a transparent gap, a denominator
    // outside the normal range) only raises the wave's flag, and the wave then redoes its segment in the reference's own form
    unsigned long long redo = 0;
    const int gc = count > 0 ? g_lo : 0;
    {
        const double2* pP = sP + gi + gc - j0;
        const double* pR = sRT + th * rstride + gc - j0;
        const double2 q0 = pP[j0], q1 = pP[j0 + 1];
        double a1 = q1.y, s1 = q1.x;
        double d10 = q0.x - s1;
        double t0 = mul_rn(q0.y * a1, pR[j0]);
#pragma unroll
        for (int j = 0; j < LMAX; ++j) {
            c[j] = 1.0, e[j] = 0.0;
            if (j >= j0) {
                if (j < LMAX - 1 || seg < NS - 1) {  // :208-249
                    const double2 q2 = pP[j + 2];
                    const double s2 = q2.x, a2 = q2.y;
                    const double t1 = mul_rn(a1 * a2, pR[j + 1]);
                    const double d21 = s2 - s1;
                    redo |= rt_coef_fast<false>(t0, t1, d10, d21, s1, c[j], e[j]);
                    t0 = t1, d10 = -d21, s1 = s2, a1 = a2;
                } else {  // the final gap (:253-266)
                    redo |= rt_coef_fast<true>(t0, 0.0, d10, 0.0, s1, c[j], e[j]);
                }
                A *= c[j];
                B = fma(c[j], B, e[j]);
            }
        }
    }
    if (redo) {  // rare: a rolled loop, the coefficients find their registers through a (scalar) switch
        A = 1.0, B = 0.0;
        for (int j = j0; j < LMAX; ++j) {
            const int gap = gc + j - j0;
            const double s0 = sP[gi + gap].x, s1 = sP[gi + gap + 1].x;
            const double t0 = mul_rn(sP[gi + gap].y * sP[gi + gap + 1].y, sRT[th * rstride + gap]);
            double cj, ej;
            if (gap < n_gap - 1) {
                const double t1 = mul_rn(sP[gi + gap + 1].y * sP[gi + gap + 2].y, sRT[th * rstride + gap + 1]);
                rt_coef_reference<false>(t0, t1, s0 - s1, sP[gi + gap + 2].x - s1, s1, cj, ej);
            } else {
                rt_coef_reference<true>(t0, 0.0, s0 - s1, 0.0, s1, cj, ej);
            }
            A *= cj;
            B = fma(cj, B, ej);
#pragma unroll
            for (int k = 0; k < LMAX; ++k)
                if (k == j) c[k] = cj, e[k] = ej;
        }
    }
    sAB[(seg * 64 + lane) * 2] = A;
    sAB[(seg * 64 + lane) * 2 + 1] = B;
    __syncthreads();
    // step 2: the intensity entering this segment (0 at the innermost point, np.zeros :134)
    double inten = 0.0;
    for (int k = 0; k < seg; ++k) inten = fma(sAB[(k * 64 + lane) * 2], inten, sAB[(k * 64 + lane) * 2 + 1]);
    if (seg == 0) {
        if (valid && I_nus) I_nus[(size_t)i * theta_stride + th] = 0.0;
        if (valid && g == 0 && F) F[i] = 0.0;
    }
    // step 3: replay, flux terms to LDS (the staging arrays are dead since the barrier)
    double* fx = sFx + (size_t)seg * LMAX * gpw * G + (active ? grp * G + g : 0);
    // (one pointer stepped per gap: left as an index expression, the 64-bit address of the optional intensity output is
    // formed for every gap whether or not it is requested — ten instructions next to one FMA)
    double* ip = valid && I_nus ? I_nus + ((size_t)(g_lo + 1) * n_nu + i) * theta_stride + th : nullptr;
    const size_t istep = (size_t)n_nu * theta_stride;
#pragma unroll
    for (int j = 0; j < LMAX; ++j) {
        if (j >= j0) {
            inten = fma(c[j], inten, e[j]);
            if (ip) *ip = inten, ip += istep;
            if (active) fx[(j - j0) * gpw * G] = inten * wt;
        }
    }
    fx = sFx + (size_t)seg * LMAX * gpw * G;
    wave_sync();
    if (F) {
        const int half = (n_theta + 1) >> 1;
        for (int p = lane; p < 2 * count * gpw; p += 64) {
            const int h = p & 1, q = p >> 1;  // q = b * gpw + gq
            const int b = (int)(((float)q + 0.5f) * (1.0f / (float)gpw)), gq = q - b * gpw;  // q < 64 LMAX: exact
            const double* cc = fx + q * G + (h ? half : 0);
            const int cnt = h ? n_theta - half : half;
            double sum = 0.0;
            for (int t = 0; t < cnt; ++t) sum = add_rn(sum, cc[t]);
            const double other = __shfl_xor(sum, 1);
            const int64_t iq = i0 + gq;
            if (h == 0 && iq < n_nu) F[(size_t)(g_lo + b + 1) * fld + iq] = add_rn(sum, other);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// scipy.ndimage.convolve1d(in, w) with the default mode='reflect' (d c b a | a b c d | d c b a), odd kernel, origin 0:
// what rotation_broadening applies to the spectrum (broadening.py:869-871).  scipy's summation order is kept:
// for a symmetric kernel  out = in[0] w[c], then pairs (in[-j] + in[+j]) w[c-j] from the OUTERMOST inwards;
// otherwise the last tap first, then the ascending-offset sum with the reversed kernel.
__device__ __forceinline__ int64_t reflect_index(int64_t i, int64_t n)
{
    const int64_t p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - 1 - i : i;
}
__global__ __launch_bounds__(kBlock) void k_convolve1d_reflect(int64_t n, const double* __restrict__ in, int m,
                                                               const double* __restrict__ w, int symmetric,
                                                               double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int h = m / 2;
    double acc;
    if (symmetric) {
        acc = mul_rn(in[i], w[h]);
        for (int j = h; j >= 1; --j) acc = add_rn(acc, mul_rn(add_rn(in[reflect_index(i - j, n)], in[reflect_index(i + j, n)]), w[h - j]));
    } else {
        acc = mul_rn(in[reflect_index(i + h, n)], w[0]);  // scipy starts from the last tap, then ascends
        for (int jj = -h; jj < h; ++jj) acc = add_rn(acc, mul_rn(in[reflect_index(i + jj, n)], w[h - jj]));
    }
    out[i] = acc;
}

}  // namespace sdx
