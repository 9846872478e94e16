/*
 * stardis_hip.h — C ABI of libstardis_hip.so: the STARDIS radiation-field hot path on MI355X (gfx950).
 *
 * The reference (tardis-sn/stardis) has no FFI or plugin registry: its hot path sits behind plain
 * Python callables whose numeric seam is "contiguous float64 ndarrays in, ndarray out" into numba
 * kernels.  Each entry point below replaces one of those numba kernels / array functions; the
 * citation on each is the reference interface it stands in for (paths relative to stardis/).
 * INTEGRATION.md shows the ctypes binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - All arithmetic is IEEE fp64.  Arrays are C-order.  int64_t sizes for the frequency / line axes.
 *   - Functions ending in _dev take DEVICE pointers and enqueue work on the context's stream
 *     (asynchronous; call sdx_synchronize or use stream order).  Functions ending in _f64 take HOST
 *     pointers, copy in, run, copy out and return after the result is on the host (what a numpy
 *     caller binds).
 *   - Return value: 0 = ok, <0 = error (SDX_ERR_*); text via sdx_last_error_string() (thread-local).
 *   - The library never keeps a caller pointer past the call.  Device scratch is owned by the context.
 *   - A context is bound to one device and one stream; use one context per host thread.
 *
 * Environment
 *   The library's results do not depend on the environment.  Two variables are read unconditionally, neither changes a
 *   computed number: SDX_RCCL_LIB (path of the RCCL library opened by sdx_group_create) and SDX_EXPERIMENT.  Everything
 *   else is an experiment / analysis knob that is honoured ONLY when SDX_EXPERIMENT=1 is set as well (tests/test_gpu_round4.py
 *   runs a synthesis under a hostile environment without the switch and compares bits):
 *     changes kernel choice or summation order (last bits of a result; shards of one spectrum must agree on them):
 *       SDX_WIDE_BLOCKS   target number of wide-role workgroups -> line subsets per (depth, tile)
 *       SDX_RT_SEG        0 / 1: never / always the segmented formal-solution kernel (the option "segmented_raytrace" wins)
 *       SDX_FAR           0 / 1: never / always the far field of the line kernels (the option "far_field" wins); SDX_FAR_RF 1 / 2 / 4
 *                         (tiles per unit of the far role / 4: scheduling only); SDX_FAR_LAUNCH: the far field as a launch of its own
 *                         (k_line_far, 8 line subsets — SDX_FAR_SPLIT 1 .. 8 — instead of the line kernel's: the order of a sum)
 *       SDX_RT_NS         4: segmented kernel with 4 waves x 14 gaps instead of 8 x 7
 *       SDX_RT_P          1, 2, 4: angles per lane of k_raytrace
 *       SDX_R_MIXED       4 / 8: grid points per lane of a mixed-precision tile
 *       SDX_NO_NARROW_SUBSETS, SDX_NARROW_SUBSETS_DENSITY (halves of a line per grid point from which the narrow role of a long
 *                         list splits its candidate lines over the four waves of a workgroup; default 8 = four lines per point)
 *     scheduling and layout only (same bits): SDX_NARROW_F (1, 2, 4 frequencies per narrow wave), SDX_NARROW_ORDER,
 *       SDX_WIDE_GROUP, SDX_CONT_DGS, SDX_CLS_BLOCKS (workgroups of a shard's classification stream), SDX_NO_CULL,
 *       SDX_NO_CONT_RIDE, SDX_NO_PREPASS_TICKET, SDX_NO_PREPASS_FRONT, SDX_PRE_LINES (32 / 48 lines per pre-pass block of a
 *       culled shard), SDX_NO_HSCAN, SDX_NO_PINNED_STAGING,
 *       SDX_SPLIT_LAUNCHES (the two roles of the line kernel as two launches, for profiling)
 *     test hook: SDX_GROUP_LOOPBACK (see sdx_group_create).
 */
#ifndef STARDIS_HIP_H
#define STARDIS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDX_OK 0
#define SDX_ERR_ARG (-1)  /* bad argument: null pointer, non-monotone grid, zero Doppler width ... */
#define SDX_ERR_HIP (-2)  /* HIP runtime error (no device, launch failure, ...) */
#define SDX_ERR_COMM (-3) /* RCCL error: library not loadable, communicator set-up or the collective itself (sdx_group_*) */
#define SDX_ERR_OOM (-4)  /* device or host allocation failed */
#define SDX_ERR_STALE (-5) /* sdx_graph_launch: the context's scratch was reallocated after the graph was captured; capture again */

/* broadening flags (broadening.py:688-691: which terms the config lists) */
#define SDX_LINEAR_STARK 1
#define SDX_QUADRATIC_STARK 2
#define SDX_VAN_DER_WAALS 4
#define SDX_RADIATION 8

typedef struct sdx_ctx sdx_ctx;

/* ---- runtime -------------------------------------------------------------------------------- */
const char* sdx_version(void);
const char* sdx_last_error_string(void);
int sdx_last_error_code(void); /* the SDX_ERR_* behind the last message (for entry points that return a pointer) */
int sdx_device_count(void); /* number of visible HIP devices, 0 when none (never an error) */
/* hipSetDevice for the calling thread: for callers that allocate device memory of their own next to a context (every sdx_* entry
 * point that takes a context selects the context's device itself where it matters) */
int sdx_set_device(int device);

/* One context per device.  stream = NULL: the context creates its own non-blocking stream;
 * otherwise an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream). */
sdx_ctx* sdx_create(int device, void* stream);
void sdx_destroy(sdx_ctx* ctx);
int sdx_set_stream(sdx_ctx* ctx, void* stream);
void* sdx_get_stream(sdx_ctx* ctx);
int sdx_synchronize(sdx_ctx* ctx);
/* options:
 *   "indexed_min_lines" (default 8192): line lists at least this long are not scanned completely by every tile of the wide
 *       role: lines whose widest window exceeds 4096 grid points are listed once and visited by every tile, all others are
 *       found by centre range (the list is sorted); frequency shards of such lists also run a culled pre-pass.
 *   "prepass_ticket_min_blocks" (default 16384): frequency shards of long lists — from this many 32-line blocks on (5e5 lines) the
 *       culled pre-pass runs as many workgroups as the chip holds, drawing the blocks that have work from a counter, instead of one
 *       workgroup per candidate block (most of which return at once).  Scheduling only: the same bits.
 *   "mixed_precision" (default 0): 1 selects the fp32-mixed TOLERANCE path (BASELINE config 5).  Wide windows: far-wing
 *       evaluations — grid tiles wholly inside a line's window and wholly in Faddeeva region I, and window edges — compute
 *       the rational y (q + y^2 + 1/2) / ((q - y^2 - 1/2)^2 + 4 q y^2) in packed fp32, two grid points per instruction:
 *       x = (nu_i - nu_l) / doppler from hi + lo fp32 splits of both frequencies (relative error ~1e-7 whatever their
 *       distance), fp32 constants per (line, depth) written by the pre-pass, v_rcp_f32, fp32 running sums flushed into the
 *       fp64 sums every >= 48 terms.  Narrow windows (half-width <= 64 grid points) and the line cores delegated to them:
 *       all four Humlicek regions in fp32 with packed complex arithmetic and the hardware exp / cos
 *       (sdx_voigt_term_f32_dev exposes that routine).  Line cores kept by the wide windows, the continuum and the formal
 *       solution stay fp64.  Stated tolerance: 1e-4 relative on the emergent flux (measured: 7e-6 on the line opacity, 7e-6
 *       of Re w per evaluation; tests/test_gpu_configs.py, tests/test_gpu_hot_faddeeva.py).  fp64 remains the default and
 *       the parity path.  Where the reference divides by zero — an opaque gap with a TRANSPARENT gap ahead of it on the ray
 *       (radiation_field_solvers/base.py:208-249 with tau[gap + 1] = 0) — the fp64 kernels reproduce its NaN; the fp32 formal
 *       solution of this mode takes that step to first order and stays finite.
 *   "segmented_raytrace" (default -1): which formal-solution kernel runs.  -1: decided from the size of the GLOBAL grid against
 *       a fixed constant (grids under 3 x 4 x 256 k_raytrace waves take the segmented kernel) — never from the shard's own
 *       width or the device's CU count, so that a frequency shard and the unsharded grid run the same arithmetic and stay
 *       bit-identical; 0: never the segmented kernel; 1: whenever it supports the shape.  The fp32-mixed twins (*_f32mix) that
 *       SURVEY §8b proposed exist for the host-buffer entry points (below) and are this library's "mixed_precision" option for the rest.
 *   "far_field" (default -1): the far field of the line opacity.  A (line, depth) item whose window (base.py:561-575) contains a whole
 *       256-point tile of the GLOBAL grid, clear of the line's core range (every point of the tile in Faddeeva region I) and with the
 *       line's centre at least three tile widths (6 half-widths, measured in frequency) from the tile's centre, is not evaluated at
 *       the tile's 256 grid points but at its 16 Chebyshev nodes — the same region-I rational, fp64 — and the summed node values are
 *       carried to the grid points by the degree-15 interpolant (the far role of the line kernel's launch; a third partial plane).  Nine tenths of the window
 *       evaluations of the 3000 - 10000 A workloads are such triples.  Interpolation error <= 11.9^-16 = 6e-18 of an item's value;
 *       measured against the direct sum: <= 7e-15 relative on the line opacity, 7e-13 on the flux (tolerances 1e-12 / 1e-10).
 *       Which triples are far is a property of the grid and the list, not of the shard: shards stay bit-identical to the unsharded
 *       run.  -1: grids of at least 32768 frequencies (decided from the GLOBAL grid); 0: never — every window point is evaluated
 *       where it lies, as the reference does; 1: whenever the line kernel runs 256-point tiles (always, unless an experiment knob
 *       says otherwise).  With "mixed_precision" = 1 the far field and what it leaves to the wide windows (a line's near zone, window
 *       edges) are evaluated in fp64 as in the fp64 mode — the far wings that mode computed in fp32 are the far field's now —; narrow
 *       windows, delegated cores and the formal solution stay fp32. 
 *   "narrow_records" (default -1): where the narrow role of the line kernel (windows of at most 64 points, delegated line cores) finds 1 / dw, y
 *       and the amplitude of a (line, depth) item.  1: in records the pre-pass writes (24 bytes per narrow item); 0: it forms them itself from
 *       the caller's doppler widths, gammas and alphas — the same three operations, so the results are bit-identical — and the pre-pass
 *       writes one byte per item (long dense fp64 lists only: lists of at least "indexed_min_lines" lines, no line-list scalars, no
 *       "mixed_precision"); -1: 0 for lists of at least four lines per grid point (1e6 lines: 1.35 GB less traffic and scratch per
 *       synthesis, the step 0.5 % faster), 1 otherwise (1.5e5 lines: the step is 0.4 % slower without the records). */
int sdx_set_int_option(sdx_ctx* ctx, const char* name, int64_t value);
/* The far-field rule as the library applies it — for planners that weigh shards (stardis_amd.parallel.column_cost) and must not
 * carry constants of their own.  sdx_far_field_active: 1 when a synthesis of a GLOBAL grid of n_nu_global points on this context runs
 * with the far field (the "far_field" option, else the automatic rule: grids of at least *min_points), 0 when not, negative on a null
 * context.  sdx_far_field_rule: the rule's constants — the automatic threshold, the tile the far field lives on (grid points) and how
 * close to its centre, in grid points, a window is still evaluated point by point (half a tile + the pole-distance ratio in tile
 * half-widths).  Replaces nothing in the reference (its sum is always the direct one, opacities_solvers/base.py:577-592). */
int sdx_far_field_active(const sdx_ctx* ctx, int64_t n_nu_global);
int sdx_far_field_rule(int64_t* min_points, int* tile_points, int* near_points);

/* device memory for callers that do not bring their own (numpy-only users) */
/* (freed blocks are kept by the context — up to 4 GiB — and handed out again: all uses are ordered on the context's stream, so the
 * Python mirror's ~30 small uploads per drop-in call cost no hipMalloc / hipFree after the first call) */
void* sdx_malloc(sdx_ctx* ctx, size_t bytes);
int sdx_free(sdx_ctx* ctx, void* ptr);
int sdx_memcpy_h2d(sdx_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes); /* async on ctx stream */
int sdx_memcpy_d2h(sdx_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes); /* synchronises */
int sdx_memset(sdx_ctx* ctx, void* dst_dev, int value, size_t bytes);
/* Page-locked host memory (hipHostMalloc) for callers that keep staging / result buffers of their own, and copies that move it
 * by DMA directly — sdx_memcpy_h2d / _d2h above take pageable pointers and go through a bounce buffer of the context.
 * sdx_memcpy_h2d_pinned is asynchronous on the context's stream: the source must stay untouched until a synchronising call
 * (sdx_memcpy_d2h[_pinned], sdx_synchronize) has returned.  sdx_host_alloc returns NULL on failure (-4 in sdx_last_error_code). */
void* sdx_host_alloc(sdx_ctx* ctx, size_t bytes);
int sdx_host_free(void* ptr);
int sdx_memcpy_h2d_pinned(sdx_ctx* ctx, void* dst_dev, const void* src_pinned, size_t bytes);
int sdx_memcpy_d2h_pinned(sdx_ctx* ctx, void* dst_pinned, const void* src_dev, size_t bytes); /* synchronises */

/* Scratch the line-opacity call needs for n_lines x n_depth; call once before stream capture. */
int sdx_reserve_line_workspace(sdx_ctx* ctx, int n_depth, int64_t n_lines);

/* stream capture -> hipGraph, so a launch-bound sequence of *_dev calls replays as one submit.  A graph bakes in the
 * context's scratch pointers: if a later, larger call on the same context makes the library reallocate scratch, launching
 * the old graph returns SDX_ERR_STALE instead of touching freed memory (reserve the largest workspace first, or re-capture). */
int sdx_graph_begin(sdx_ctx* ctx);
int sdx_graph_end(sdx_ctx* ctx, void** graph_exec_out);
int sdx_graph_launch(sdx_ctx* ctx, void* graph_exec);
int sdx_graph_destroy(sdx_ctx* ctx, void* graph_exec);

/* HIP-event timing on the context's stream (bench.py: whole region and per-kernel durations) */
int sdx_timer_start(sdx_ctx* ctx);
int sdx_timer_stop(sdx_ctx* ctx, double* elapsed_ms);
int sdx_profile_enable(sdx_ctx* ctx, int on); /* bracket every kernel launch with events */
int sdx_profile_reset(sdx_ctx* ctx);
int sdx_profile_get(sdx_ctx* ctx, const char* kernel, int64_t* launches, double* total_ms);
/* which device kernel last ran under the stage name `kernel` while profiling was on ("k_raytrace" -> "k_raytrace_seg<8,7>",
 * "k_raytrace<1>", "k_raytrace_f32", ...); "" when the stage has one kernel only or did not run */
int sdx_profile_variant(sdx_ctx* ctx, const char* kernel, char* out, int out_len);

/* ---- line opacity ---------------------------------------------------------------------------
 * Replaces calc_alan_entries (radiation_field/opacities/opacities_solvers/base.py:487-592):
 *   out[d, i] (+)= sum_l [lo_ld <= i < hi_ld] alpha_ld * voigt(nu_i - nu_l; doppler_ld, gamma_ld)
 * with the reference's window rule (:556-575) evaluated on the GLOBAL grid `nus` (n_nu, descending).
 * Only columns nu_begin .. nu_begin+nu_count-1 are produced (frequency sharding); out is
 * [n_depth][out_ld] with column 0 = global index nu_begin.  line_nus ascending; doppler/alphas are
 * [n_lines][n_depth]; gammas is [n_lines][gamma_cols], gamma_cols = n_depth or 1 (:547-551).
 * accumulate = 0 overwrites out, 1 adds to it.  n_evaluations_dev (optional, device int64)
 * receives sum(hi - lo) over all (line, depth), i.e. the number of Voigt evaluations of the full grid (0 for an empty list; a call
 * with nu_count = 0 returns at once and leaves it untouched — sdx_synthesize_sharded_f64 asks its first NON-empty rank). */
int sdx_line_opacity_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t nu_begin,
                         int64_t nu_count, int64_t n_lines, const double* line_nus, const double* doppler_widths,
                         const double* gammas, int gamma_cols, const double* alphas, double* out, int64_t out_ld,
                         int accumulate, int64_t* n_evaluations_dev);
int sdx_line_opacity_f64(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t n_lines,
                         const double* line_nus, const double* doppler_widths, const double* gammas, int gamma_cols,
                         const double* alphas, double* out, int64_t* n_evaluations);
/* window rule only (:556-575): lower/upper are [n_lines][n_depth] int32, on device */
int sdx_line_windows_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t n_lines,
                         const double* line_nus, const double* doppler_widths, const double* gammas, int gamma_cols,
                         const double* alphas, int32_t* lower, int32_t* upper);

/* voigt.py:89-91 faddeeva (z, w interleaved re/im) and voigt.py:153-155 voigt_profile, element-wise */
int sdx_faddeeva_dev(sdx_ctx* ctx, int64_t n, const double* z, double* w);
int sdx_voigt_profile_dev(sdx_ctx* ctx, int64_t n, const double* delta_nu, const double* doppler_width,
                          const double* gamma, double* phi);
/* The hot-path variant of the same function, element-wise: what the line kernels evaluate per (line, depth, frequency)
 * from the pre-pass constants — out = amp * Re w(delta_nu * inv_doppler_width + i y), real part only, FMA arithmetic
 * (voigt.py:17-86 regions and predicates; base.py:627 amplitude).  A test hook: it lets the kernel's own routine be
 * compared point by point with the reference's Faddeeva / Voigt vectors. */
int sdx_voigt_term_dev(sdx_ctx* ctx, int64_t n, const double* delta_nu, const double* inv_doppler_width, const double* y,
                       const double* amp, double* out);
/* The fp32 routine the narrow windows are evaluated with when the option mixed_precision is on (all four regions in packed
 * fp32, hardware exp / cos); same arguments, rounded to fp32 inside.  Test hook for the stated 1e-4 tolerance of that mode. */
int sdx_voigt_term_f32_dev(sdx_ctx* ctx, int64_t n, const double* delta_nu, const double* inv_doppler_width, const double* y,
                           const double* amp, double* out);

/* ---- broadening (opacities_solvers/broadening.py) -------------------------------------------
 * calc_gamma :550-656 with calculate_broadening's argument preparation :706-721 (ion_number is the
 * reference's ion_number + 1); per-line inputs [n_lines], per-depth inputs [n_depth], out [n_lines][n_depth]. */
int sdx_calc_gamma_dev(sdx_ctx* ctx, int64_t n_lines, int n_depth, const int32_t* atomic_number,
                       const int32_t* ion_number, const double* ionization_energy, const double* upper_level_energy,
                       const double* lower_level_energy, const double* A_ul, const double* electron_density,
                       const double* temperature, const double* h_density, int flags, double* gammas);
/* calc_doppler_width :32-71 broadcast as at :723-730 / :814-819 */
int sdx_doppler_widths_dev(sdx_ctx* ctx, int64_t n_lines, int n_depth, const double* line_nus, const double* mass,
                           const double* temperature, double microturbulence, double* doppler_widths);
/* calc_vald_gamma :1009-1085 (stark :880-890, van der Waals :893-1006).  flags: 1 linear Stark (hydrogen lines), 2 VALD Stark,
 * 4 van der Waals, 8 radiation; + 16 = the sum is NOT halved: the VALD branch of calculate_molecule_broadening (:771-799), which adds
 * A_ul, calc_vald_stark_gamma when linear OR quadratic Stark is configured (pass 2, never 1) and calc_vald_vdW and skips :1084. */
int sdx_calc_vald_gamma_dev(sdx_ctx* ctx, int64_t n_lines, int n_depth, const int32_t* atomic_number,
                            const int32_t* ion_number, const double* ionization_energy,
                            const double* upper_level_energy, const double* lower_level_energy, const double* A_ul,
                            const double* stark, const double* waals, const double* mass,
                            const double* electron_density, const double* temperature, const double* h_density,
                            int flags, double* gammas);

/* The reference's element-wise ufuncs, all operands already broadcast to length n (integers passed as doubles):
 *   op 0 calc_doppler_width(nu_line, temperature, atomic_mass, microturbulence)                    :69-71
 *   op 1 calc_n_effective(ion_number, ionization_energy, level_energy)                            :140-146
 *   op 2 calc_gamma_linear_stark(n_eff_upper, n_eff_lower, electron_density)                      :232-234
 *   op 3 calc_gamma_quadratic_stark(ion_number, n_eff_upper, n_eff_lower, electron_density, T)    :346-360
 *   op 4 calc_gamma_van_der_waals(ion_number, n_eff_upper, n_eff_lower, T, h_density)             :476-490 */
int sdx_broadening_scalar_dev(sdx_ctx* ctx, int op, int64_t n, const double* a, const double* b, const double* c,
                              const double* d, const double* e, double* out);

/* ---- continuum (opacities_solvers/base.py:40-317); every out is [n_depth][ld], columns nu_begin.. -- */
/* calc_alpha_file :40-70 with a 1-D table (util.py:94-103, np.interp): out = sigma(lambda_i) * density[d] */
int sdx_alpha_file_1d_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* lambdas, int n_table,
                          const double* table_wavelength, const double* table_sigma, const double* density,
                          double* out, int64_t ld);
/* calc_alpha_file :70 with sigma already tabulated per (depth, nu) (util.py:35-91 host interpolation) */
int sdx_alpha_file_2d_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* sigma, int64_t sigma_ld,
                          const double* density, double* out, int64_t ld);
/* sigma_file for the two 2-D tables (opacities_solvers/util.py:35-91): scipy's
 * LinearNDInterpolator(points, values, fill_value=0) on the Delaunay triangulation of the rectilinear table, evaluated at
 * the mesh (lambdas[k], second[d]) with second = T (H2+ bf) or 5040/T (H- ff).  The caller passes the triangulation its
 * scipy/Qhull built, indexed by grid cell: cell_simplices[(i*(n_y-1)+j)*2 + {0,1}] are the two triangles of cell (i, j),
 * transform[s] is Qhull's 3x2 barycentric transform of triangle s (scipy.spatial.Delaunay.transform),
 * simplex_values[s][v] the table value at its v-th vertex.  scale_kind 0: raw; 1: x 1e-18 (util.py:58);
 * 2: x 1e-26 x k_B x T[d] (util.py:83-88).  zero_rows (optional, [n_depth]) is set to 1 for rows that contain an exact
 * zero — the rows the reference names in its "outside of interpolation range" warning.  sigma is [n_depth][ld]. */
int sdx_sigma_table_2d_dev(sdx_ctx* ctx, int n_x, const double* x_axis, int n_y, const double* y_axis,
                           const int32_t* cell_simplices, const double* transform, const double* simplex_values,
                           int n_depth, int64_t n_nu, const double* lambdas, const double* second, int scale_kind,
                           const double* temperature, double* sigma, int64_t ld, int32_t* zero_rows);
/* calc_alpha_bf :178-271: levels grouped by species; cutoff[L] = (E_ion - E_exc)/h; level_density [n_levels][n_depth] */
int sdx_alpha_bf_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int n_species,
                     const int32_t* species_offsets, const int32_t* species_ion_number, const double* cutoff,
                     const double* level_density, double* out, int64_t ld);
/* calc_alpha_ff :274-317: number_density [n_species][n_depth] = n_ion * n_e, ion_number as get_number_density returns */
int sdx_alpha_ff_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, const double* temperature,
                     int n_species, const int32_t* species_ion_number, const double* number_density, double* out,
                     int64_t ld);
/* calc_alpha_rayleigh :74-135; NULL density = species not requested.  `nus` is modified in place
 * (nu > 2.3e15 -> 0) exactly as the reference does at :99. */
int sdx_alpha_rayleigh_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, double* nus, const double* n_h,
                           const double* n_he, const double* n_h2, double* out, int64_t ld);
/* calc_alpha_electron :139-174 */
int sdx_alpha_electron_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* electron_density, double* out,
                           int64_t ld);
/* Opacities.calc_total_alphas (radiation_field/opacities/base.py:24-28): total += src */
int sdx_accumulate_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, double* total, int64_t total_ld, const double* src,
                       int64_t src_ld);

/* ---- formal solution (radiation_field/radiation_field_solvers/base.py) ---------------------- */
/* blackbody_flux_at_nu (source_functions/blackbody.py:10-35) */
int sdx_blackbody_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, const double* temperature,
                      double* out, int64_t ld);
/* calc_weights_parallel :6-47 */
int sdx_calc_weights_dev(sdx_ctx* ctx, int64_t n, const double* delta_tau, double* w0, double* w1, double* w2);
/* raytrace :271-346, plane-parallel branch.  ray_dist is [n_depth-1][n_theta] = dist[:,None]/cos(thetas)
 * (:302-305).  nus/total_alphas/F_nu hold the n_nu columns being traced (a shard passes its own
 * slice).  accumulate = 1: F_nu is ACCUMULATED into, the reference's behaviour (:336); 0: F_nu is overwritten
 * (row 0 becomes 0).  I_nus [n_depth][n_nu][n_theta] optional (:333-334). */
int sdx_raytrace_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, int n_theta, const double* nus,
                     const double* temperature, const double* ray_dist, const double* theta_weights,
                     const double* total_alphas, int64_t alpha_ld, double* F_nu, int64_t F_ld, double* I_nus,
                     int accumulate);
/* raytrace :271-346, spherical branch: ray_dist from calculate_spherical_ray (:349-381, zeros where a ray misses a
 * shell), the inward sweep of single_theta_trace_parallel (:141-198) before the outward one, and finally
 * F_nu *= photospheric_correction = (r[-1] / reference_r)**2 (:340-344). */
int sdx_raytrace_spherical_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, int n_theta, const double* nus,
                               const double* temperature, const double* ray_dist, const double* theta_weights,
                               const double* total_alphas, int64_t alpha_ld, double* F_nu, int64_t F_ld, double* I_nus,
                               int accumulate, double photospheric_correction);
/* raytrace with the source function handed over as a plane [n_depth][source_ld] instead of the Planck function the other
 * entry points evaluate in the kernel: RadiationField.source_function is any callable (nu, T[:, None]) -> (N_d, N_nu)
 * (radiation_field/base.py:12-68, used at radiation_field_solvers/base.py:133); the host evaluates a foreign one and uploads
 * the result.  source = NULL: Planck.  inward != 0: spherical geometry (as sdx_raytrace_spherical_dev, with the correction). */
int sdx_raytrace_source_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, int n_theta, const double* nus,
                            const double* temperature, const double* ray_dist, const double* theta_weights,
                            const double* total_alphas, int64_t alpha_ld, const double* source, int64_t source_ld,
                            double* F_nu, int64_t F_ld, double* I_nus, int accumulate, int inward,
                            double photospheric_correction);
int sdx_raytrace_f64(sdx_ctx* ctx, int n_depth, int64_t n_nu, int n_theta, const double* nus,
                     const double* temperature, const double* ray_dist, const double* theta_weights,
                     const double* total_alphas, double* F_nu, double* I_nus);

/* ---- fused synthesis for resident data (the benchmark path) ---------------------------------
 * total_alphas[d,i] = ((((file + bf) + ff) + rayleigh) + electron) + line   in calc_alphas order
 * (:655-738), then raytrace.  Any source pointer group may be NULL (skipped, contributes nothing).
 * All arrays are device-resident; outputs cover columns nu_begin .. nu_begin+nu_count-1. */
typedef struct sdx_continuum {
    /* alpha_file_Hminus_bf style 1-D table source */
    const double* lambdas;          /* [n_nu] Angstrom, global grid */
    int n_table;
    const double* table_wavelength; /* [n_table] */
    const double* table_sigma;      /* [n_table] */
    const double* table_density;    /* [n_depth] */
    /* bf */
    int bf_n_species;
    int bf_n_levels;                /* = bf_species_offsets[bf_n_species], known to the host */
    const int32_t* bf_species_offsets;
    const int32_t* bf_species_ion_number;
    const double* bf_cutoff;
    const double* bf_level_density;
    /* ff */
    int ff_n_species;
    const int32_t* ff_species_ion_number;
    const double* ff_number_density;
    /* rayleigh (all NULL: source returns zeros, as the reference does for an empty species list) */
    const double* ray_n_h;
    const double* ray_n_he;
    const double* ray_n_h2;
    int rayleigh_enabled;
    /* electron */
    const double* electron_density; /* NULL = disable_electron_scattering */
    const double* temperature;      /* [n_depth] */
    /* further alpha_file_<source> planes, already formed on the device (sdx_alpha_file_1d_dev / sdx_alpha_file_2d_dev: the
     * two-dimensional tables Hminus_ff and H2plus_bf, or several tabulated sources at once): [n_depth][file_plane_ld] each,
     * column 0 = grid index 0, added right after the 1-D table above in this order — with the table left out (table_sigma NULL)
     * the planes are calc_alphas' file loop (:666-677) in the configuration's own order.  Device pointers only: the host-buffer
     * entry points (*_f64) refuse a description that carries planes. */
    int n_file_planes;              /* 0 .. 4 */
    const double* file_plane[4];
    int64_t file_plane_ld;
} sdx_continuum;

/* calc_alpha_file / calc_alpha_bf / calc_alpha_ff / calc_alpha_rayleigh / calc_alpha_electron (opacities_solvers/base.py:40-317)
 * and their sum in calc_alphas' order (:655-700) for a caller that holds everything in HOST memory (numpy through ctypes, C):
 * `cont` carries host pointers here (array lengths follow from the sizes in the struct; no file planes), every output is an
 * optional host array [n_depth][n_nu]; one upload of the inputs, one download per requested plane, on the context's
 * persistent staging.  alpha_rayleigh != NULL also reproduces the reference's side effect (:99): frequencies above 2.3e15 Hz
 * are set to 0 in the caller's `nus`.  alpha_electron with electron_density == NULL is a plane of zeros (the reference
 * returns the scalar 0 there, :164-165).  total_alphas = ((((file + bf) + ff) + rayleigh) + electron), sources that are
 * absent contributing nothing. */
int sdx_continuum_f64(sdx_ctx* ctx, int n_depth, int64_t n_nu, double* nus, const sdx_continuum* cont, double* alpha_file,
                      double* alpha_bf, double* alpha_ff, double* alpha_rayleigh, double* alpha_electron, double* total_alphas);

int sdx_total_alphas_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t nu_begin,
                         int64_t nu_count, const sdx_continuum* cont, const double* alpha_line, int64_t line_ld,
                         double* total_alphas, int64_t total_ld);

/* ---- post-processing ---------------------------------------------------------------------------
 * scipy.ndimage.convolve1d(in, weights) with mode='reflect', origin 0, odd kernel length: the operation
 * rotation_broadening applies to the spectrum (opacities_solvers/broadening.py:869-871).  symmetric != 0 selects
 * scipy's pairwise order for symmetric kernels (the caller applies scipy's DBL_EPSILON symmetry test). */
int sdx_convolve1d_reflect_dev(sdx_ctx* ctx, int64_t n, const double* in, int m, const double* weights, int symmetric,
                               double* out);
/* spectrum_lambda = F_nu * nu / lambda, element-wise (stardis/base.py:137-141) — keeps the emergent spectrum on the
 * device between the formal solution and the instrumental / rotational convolutions */
int sdx_flux_nu_to_lambda_dev(sdx_ctx* ctx, int64_t n, const double* f_nu, const double* nus, const double* lambdas, double* out);

/* Everything in one call for resident data: pre-pass + line opacity + total (above) + raytrace (F_nu
 * overwritten).  F_nu is [n_depth][ld].  alpha_line_out and total_alphas ([n_depth][ld]) are OPTIONAL outputs (NULL: the
 * plane is never written to HBM — the reference only reads them back through opacities_dict / Opacities.total_alphas;
 * the formal solution forms total = continuum + line while staging its columns).  This is the step bench.py times. */
int sdx_synthesize_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t nu_begin, int64_t nu_count,
                       int64_t n_lines, const double* line_nus, const double* doppler_widths, const double* gammas,
                       int gamma_cols, const double* alphas, const sdx_continuum* cont, int n_theta,
                       const double* temperature, const double* ray_dist, const double* theta_weights,
                       double* alpha_line_out, double* total_alphas, double* F_nu, int64_t ld,
                       int64_t* n_evaluations_dev);
/* The same step with the two optional extras of RadiationField (radiation_field/base.py:38-68).  source: the caller's source
 * function already evaluated, [n_depth][source_ld] with column 0 = column nu_begin (RadiationField.source_function is any callable
 * (nu, T) -> (N_d, N_nu), radiation_field_solvers/base.py:133; NULL: the Planck function, evaluated by the kernel).  I_nus:
 * track_individual_intensities (:64-68, filled at radiation_field_solvers/base.py:324-338), [n_depth][nu_count][n_theta] receives
 * the intensity of every ray at every depth point (row 0: zeros, the reference's initial condition :134); NULL: not kept. */
int sdx_synthesize_ex_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t nu_begin, int64_t nu_count,
                          int64_t n_lines, const double* line_nus, const double* doppler_widths, const double* gammas,
                          int gamma_cols, const double* alphas, const sdx_continuum* cont, int n_theta,
                          const double* temperatures, const double* ray_dist, const double* weights, double* alpha_line_out,
                          double* total_alphas, double* F_nu, int64_t ld, const double* source, int64_t source_ld, double* I_nus,
                          int64_t* n_evaluations_dev);

/* The same synthesis for a caller whose data lives in host memory (plain C, or numpy through ctypes): every pointer,
 * including those inside `cont`, is a HOST pointer; arrays are uploaded, sdx_synthesize_dev runs, results come back.
 * F_nu is [n_depth][n_nu] (overwritten); alpha_line_out, total_alphas ([n_depth][n_nu]) and n_evaluations are optional.
 * ray_dist is the (n_depth-1, n_theta) table dist[:, None] / cos(theta) (radiation_field_solvers/base.py:302-305). */
int sdx_synthesize_f64(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t n_lines, const double* line_nus,
                       const double* doppler_widths, const double* gammas, int gamma_cols, const double* alphas,
                       const sdx_continuum* cont, int n_theta, const double* temperature, const double* ray_dist,
                       const double* theta_weights, double* alpha_line_out, double* total_alphas, double* F_nu,
                       int64_t* n_evaluations);

/* ---- fp32-mixed twins of the host-buffer entry points (`*_f32mix`) ---------------------------
 * The same signatures and semantics as sdx_line_opacity_f64 / sdx_raytrace_f64 / sdx_synthesize_f64 with the context's
 * "mixed_precision" option on for the duration of the call (and restored afterwards, whatever it was): the tolerance path of BASELINE
 * configs[4] — far wings, window edges and narrow windows in packed fp32, the formal solution of plane-parallel grids in fp32; the
 * pre-pass, the continuum and kept line cores in fp64; stated tolerance 1e-4 on the flux against the fp64 path (measured 1e-6).  The
 * device-pointer entry points (*_dev) take the option instead: sdx_set_int_option(ctx, "mixed_precision", 1).  The reference has no
 * counterpart (its arithmetic is fp64 throughout, opacities_solvers/base.py:487-627, radiation_field_solvers/base.py:85-346). */
int sdx_line_opacity_f32mix(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t n_lines,
                            const double* line_nus, const double* doppler_widths, const double* gammas, int gamma_cols,
                            const double* alphas, double* out, int64_t* n_evaluations);
int sdx_raytrace_f32mix(sdx_ctx* ctx, int n_depth, int64_t n_nu, int n_theta, const double* nus,
                        const double* temperature, const double* ray_dist, const double* theta_weights,
                        const double* total_alphas, double* F_nu, double* I_nus);
int sdx_synthesize_f32mix(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t n_lines, const double* line_nus,
                          const double* doppler_widths, const double* gammas, int gamma_cols, const double* alphas,
                          const sdx_continuum* cont, int n_theta, const double* temperature, const double* ray_dist,
                          const double* theta_weights, double* alpha_line_out, double* total_alphas, double* F_nu,
                          int64_t* n_evaluations);

/* ---- one process, several GPUs ---------------------------------------------------
 * The frequency axis shards with no data-path exchange: every output column depends only on its own frequency
 * (radiation_field_solvers/base.py:200 is a prange over nu; calc_alan_entries :548-590 is a pure gather per (depth, nu)).
 * A group is one context + one stream per device and ONE RCCL communicator over them (ncclCommInitAll).
 * sdx_synthesize_sharded_f64 = sdx_synthesize_f64 with the columns split over the group's devices: the line list is
 * replicated, every device applies the window rule on the GLOBAL grid (global d_nu, centres, clamp — results equal the
 * single-GPU call bit for bit), and ONE ncclAllGather of the zero-padded F_nu[-1] shards (8 * max shard width bytes per
 * rank, over xGMI) leaves the emergent spectrum on every device; it is returned from device 0.
 *   devices      NULL = devices 0 .. n_gpus-1; each device at most once.
 *   shard_begin  NULL = equal blocks of ceil(n_nu / n_gpus) columns; else [n_gpus + 1] ascending column indices from 0 to n_nu
 *                (e.g. shards of equal estimated work).
 *   outputs      all host, all optional except that one of F_nu / emergent_flux must be given: alpha_line_out, total_alphas,
 *                F_nu are [n_depth][n_nu] (each device fills the columns it owns), emergent_flux [n_nu] = the gathered F_nu[-1].
 * RCCL is opened at run time (librccl.so.1; env SDX_RCCL_LIB overrides the path); any RCCL failure returns SDX_ERR_COMM.
 * Test hook: with SDX_EXPERIMENT=1 and SDX_GROUP_LOOPBACK=1 in the environment a group may list one device several times and the gather is done
 * with plain device copies instead of RCCL (RCCL refuses two ranks on one GPU) — for exercising the sharding logic on a
 * one-GPU box, never a product mode. */
typedef struct sdx_group sdx_group;
sdx_group* sdx_group_create(int n_gpus, const int* devices); /* NULL on error: sdx_last_error_code / _string */
void sdx_group_destroy(sdx_group* group);
int sdx_group_size(const sdx_group* group);
sdx_ctx* sdx_group_context(sdx_group* group, int rank); /* the rank's context (options, profiling); owned by the group */
/* what the last sharded call's collective was: ranks in the communicator, bytes each rank contributed, RCCL's version code
 * (0 in loop-back mode) */
int sdx_group_last_gather(const sdx_group* group, int* ranks, int64_t* bytes_per_rank, int* rccl_version);
/* n_evaluations (optional): the number of Voigt evaluations of the whole synthesis — a global figure, counted by the first rank
 * that owns columns.  Counting needs every window of every line, so that rank runs the full (unculled) pre-pass and becomes the
 * straggler of the group: a diagnostic, pass NULL in timed loops. */
int sdx_synthesize_sharded_f64(sdx_group* group, int n_depth, int64_t n_nu, const double* nus, int64_t n_lines,
                               const double* line_nus, const double* doppler_widths, const double* gammas, int gamma_cols,
                               const double* alphas, const sdx_continuum* cont, int n_theta, const double* temperature,
                               const double* ray_dist, const double* theta_weights, const int64_t* shard_begin,
                               double* alpha_line_out, double* total_alphas, double* F_nu, double* emergent_flux,
                               int64_t* n_evaluations);

/* ---- line parameters generated on the device ---------------------------------------
 * Instead of the three dense (N_l, N_d) tables the reference builds on the host before calc_alan_entries —
 * alpha_line (plasma/base.py:178-321 AlphaLineVald, :324-455 AlphaLineShortlistVald; plasma/molecules.py:192-320,
 * :322-440 the molecular twins), gammas and doppler_widths (opacities_solvers/broadening.py:659-732
 * calculate_broadening, :735-821 calculate_molecule_broadening, :1009-1085 calc_vald_gamma) — the caller passes
 * ~80 B of per-line scalars plus per-depth state, and the pre-pass evaluates the three values per (line, depth) itself.
 * Lines are sorted by ascending nu and already restricted to the grid's range, exactly as
 * calc_alpha_line_at_nu prepares them (opacities_solvers/base.py:392-421).  All pointers are device pointers.
 *   alpha = ((alpha_coefficient * ((exp(-e_low/(k T)) * pop[pop_row]) [* g_lo])) * strength) * (1 - exp(-h nu/(k T)))
 * gamma_mode: 0 = calc_gamma (broadening.py:550-656), 1 = calc_vald_gamma (:1009-1085), 2 = A_ul only, one column
 * (molecules, :799-801), 3 = zero.  broadening_flags: 1 linear Stark | 2 quadratic Stark | 4 van der Waals | 8 radiation.
 * ion_number is the charge seen by the outer electron (`ion_number + 1`, broadening.py:708-709). */
typedef struct sdx_linelist {
    int64_t n_lines;
    const double* nu;          /* [n_lines] Hz, ascending */
    const double* e_low_ev;    /* [n_lines] lower level energy, eV */
    const double* g_lo;        /* [n_lines] 2 j_lo + 1; NULL for short lists */
    const double* strength;    /* [n_lines] f_lu = 10**log_gf / g_lo, or 10**log_gf for short lists */
    const int32_t* pop_row;    /* [n_lines] row of pop for the line's ion or molecule */
    const double* pop;         /* [n_pop_rows][n_depth] number density / partition function */
    int n_pop_rows;
    double alpha_coefficient;  /* pi e^2 / (m_e c), cgs */
    const double* mass;        /* [n_lines] g */
    double microturbulence;    /* cm/s */
    int gamma_mode;
    int broadening_flags;
    const int32_t* atomic_number;
    const int32_t* ion_number;
    const double* ionization_energy; /* erg */
    const double* upper_energy;      /* erg */
    const double* lower_energy;      /* erg */
    const double* A_ul;
    const double* stark;       /* VALD log Stark parameter (gamma_mode 1) */
    const double* waals;       /* VALD van der Waals parameter (gamma_mode 1) */
    const double* temperature;       /* [n_depth] K */
    const double* electron_density;  /* [n_depth] cm^-3 */
    const double* h_density;         /* [n_depth] neutral hydrogen, cm^-3 */
} sdx_linelist;

/* The reference's dense tables from a line list (parity checks; any output may be NULL).  gammas is
 * [n_lines][n_depth], or [n_lines][1] for gamma_mode 2 and 3. */
int sdx_line_params_dev(sdx_ctx* ctx, int n_depth, const sdx_linelist* lines, double* alphas, double* gammas,
                        double* doppler_widths);

/* AlphaLine.calculate (plasma/base.py:146-175), lines from TARDIS atomic data: alphas[l][d] =
 * ((alpha_coefficient * level_density[lower_index[l]][d]) * stim[l][d]) * f_lu[l].  level_density is
 * [n_levels][n_depth]; stim (the stimulated-emission factor TARDIS supplies) and alphas are [n_lines][n_depth].
 * lower_index must lie in [0, n_levels) (numpy take(mode="raise") on the host side). */
int sdx_alpha_line_levels_dev(sdx_ctx* ctx, int64_t n_lines, int n_depth, int n_levels, const double* level_density,
                              const int32_t* lower_index, const double* stim, const double* f_lu,
                              double alpha_coefficient, double* alphas);

/* The fused step with everything optional a RadiationField configuration can ask for, in one description (the entry point the
 * drop-in create_stellar_radiation_field binds for the configurations sdx_synthesize_dev / _ex_dev do not cover):
 *   source / I_nus       as in sdx_synthesize_ex_dev;
 *   inward_rays          spherical models (radiation_field_solvers/base.py:141-198, :296-300, :340-344): ray_dist is the chord table
 *                        of calculate_spherical_ray (:349-381), the surface-to-centre sweep runs before the outward one and
 *                        F_nu is finally scaled by photospheric_correction = (r[-1] / reference_r)^2;
 *   line_plane[]         further line-opacity planes the caller has formed on the device ([n_depth][line_plane_ld], column 0 =
 *                        column nu_begin; sdx_line_opacity_dev / _linelist_dev): the molecular list of include_molecules
 *                        (opacities_solvers/base.py:444-484, :716-736), added AFTER the step's own line opacity in this order —
 *                        total = ((continuum + line) + plane 0) + plane 1, what Opacities.calc_total_alphas does with the
 *                        dictionary entries (opacities/base.py:24-28);
 *   linelist             when set, the step's own lines come as per-line scalars (f1, below) and the dense arrays
 *                        (line_nus ... alphas, n_lines) are ignored.
 * A zero-initialised description is sdx_synthesize_dev, bit for bit; `options` itself must not be NULL (-1). */
typedef struct sdx_synthesis_options {
    const double* source;
    int64_t source_ld;
    double* I_nus;
    int inward_rays;
    double photospheric_correction;
    int n_line_planes; /* 0 .. 2 */
    const double* line_plane[2];
    int64_t line_plane_ld;
    const sdx_linelist* linelist;
    const double* line_m_max; /* two-collective mode (below): the gathered per-line maxima [n_lines], or NULL */
} sdx_synthesis_options;
int sdx_synthesize_opt_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t nu_begin, int64_t nu_count,
                           int64_t n_lines, const double* line_nus, const double* doppler_widths, const double* gammas,
                           int gamma_cols, const double* alphas, const sdx_continuum* cont, int n_theta,
                           const double* temperatures, const double* ray_dist, const double* weights, double* alpha_line_out,
                           double* total_alphas, double* F_nu, int64_t ld, const sdx_synthesis_options* options,
                           int64_t* n_evaluations_dev);

/* Frequency shards of long lists in TWO collectives (optional; SURVEY §8e plans one).  A shard's step begins by classifying EVERY
 * line of the replicated list — the largest (gamma + doppler_width) alpha over the depths, from which the window rule
 * (opacities_solvers/base.py:561-575) gives the line's widest window: which lines can reach the shard from anywhere — a stream of
 * 8 n_lines (2 n_depth + gamma_cols) bytes that every rank repeats and that does not shrink with the number of ranks.  Here each
 * rank classifies a SHARE of the lines and the ranks exchange the shares (an all-gather of 8 n_lines bytes):
 *   1. sdx_synthesize_classify_dev(...)       m_max[line_begin .. line_begin + line_count) = this rank's share; the same launch
 *                                             leaves the grid spacing, the shard's line ranges and the continuum plane of the step
 *                                             in the context's scratch;
 *   2. the caller gathers m_max [n_lines]      (RCCL / torch.distributed; stardis_amd.parallel.ClassificationGatherer);
 *   3. sdx_synthesize_opt_dev(..., options->line_m_max = m_max)   the rest of the step — same context, same grid, shard, lists.
 * Results are bit-identical to the one-call step (the classes of the lines follow from the same numbers).  Only for culled
 * shards: dense lists of >= `indexed_min_lines` lines, grids of more than 16384 points, nu_count < n_nu, no evaluation count;
 * anything else is SDX_ERR_ARG (-1), as is step 3 without step 1.  What step 1 leaves in the scratch belongs to ONE step 3: it is
 * consumed by it, and invalidated by any other call on the context that prepares lines or a continuum plane in between (a second
 * step 3, a synthesis of another problem, sdx_line_opacity_dev ...: SDX_ERR_ARG instead of another problem's data).  A recorded
 * hipGraph replays steps 1 and 3 without these host-side checks: replay them in pairs.  The "far_field" option may differ between
 * the steps (step 3 computes the tiles' far ranges itself when step 1 has not). */
int sdx_synthesize_classify_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t nu_begin, int64_t nu_count,
                                int64_t n_lines, const double* line_nus, const double* doppler_widths, const double* gammas,
                                int gamma_cols, const double* alphas, const sdx_continuum* cont, int64_t line_begin,
                                int64_t line_count, double* m_max);

/* sdx_line_opacity_dev / sdx_synthesize_dev with the line parameters generated in the pre-pass. */
int sdx_line_opacity_linelist_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t nu_begin,
                                  int64_t nu_count, const sdx_linelist* lines, double* out, int64_t out_ld, int accumulate,
                                  int64_t* n_evaluations_dev);
int sdx_synthesize_linelist_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t nu_begin,
                                int64_t nu_count, const sdx_linelist* lines, const sdx_continuum* cont, int n_theta,
                                const double* temperature, const double* ray_dist, const double* theta_weights,
                                double* alpha_line_out, double* total_alphas, double* F_nu, int64_t ld,
                                int64_t* n_evaluations_dev);

#ifdef __cplusplus
}
#endif
#endif /* STARDIS_HIP_H */
