// stardis_hip.hip — C ABI (include/stardis_hip.h) over the gfx950 kernels in sdx_kernels.h.
// Build: stardis_amd/csrc/Makefile  (hipcc --offload-arch=gfx950 -O3 -ffp-contract=off)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only: the library is opened at run time (RcclApi below)

#include <memory>
#include <thread>

#include "../../include/stardis_hip.h"
#include "sdx_kernels.h"

using namespace sdx;

namespace {

thread_local std::string g_error;
thread_local int g_error_code = 0;

// Experiment and analysis knobs (SDX_RT_SEG, SDX_WIDE_BLOCKS, SDX_SPLIT_LAUNCHES, ... — listed in include/stardis_hip.h): some of
// them change the order of summation or the choice of kernel, i.e. the last bits of a result and the "union of shards ==
// single GPU, bit for bit" invariant when ranks inherit different environments.  They are read ONLY when the one documented
// switch SDX_EXPERIMENT=1 is set; without it the environment cannot change what the library computes.
const char* knob(const char* name)
{
    static const bool on = [] {
        const char* e = std::getenv("SDX_EXPERIMENT");
        return e && std::atoi(e) == 1;
    }();
    return on ? std::getenv(name) : nullptr;
}

int fail(int code, const std::string& msg)
{
    g_error = msg;
    g_error_code = code;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? SDX_ERR_OOM : SDX_ERR_HIP,                     \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                        \
    } while (0)

#define REQUIRE(cond, msg)                                  \
    do {                                                    \
        if (!(cond)) return fail(SDX_ERR_ARG, msg);         \
    } while (0)

struct ProfileRecord {
    const char* name;
    const char* variant;  // which device kernel ran under this name (k_raytrace: "k_raytrace<1>", "k_raytrace_seg<8,7>", ...), or nullptr
    hipEvent_t start, stop;
};

}  // namespace

struct sdx_ctx {
    int device = 0;
    int n_cu = 256;  // compute units of the device
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // line-opacity scratch (depth-major pre-pass arrays)
    void* line_ws = nullptr;
    size_t line_ws_bytes = 0;
    // small scratch: d_nu partials, evaluation counter, bf coefficients
    void* small_ws = nullptr;
    size_t small_ws_bytes = 0;
    // partial line-opacity planes [n_split + 1][n_depth][nu_count] (last plane: narrow windows)
    void* part_ws = nullptr;
    size_t part_ws_bytes = 0;
    void* far_ws = nullptr;  // far_range of the line kernels' far field: two ints per global tile
    size_t far_ws_bytes = 0;
    FarReq far_req{nullptr, 0, 0};  // the tiles whose ranges the step's next grid-spacing launch computes on the side (count = 0: none)
    bool far_req_done = false;       // ... and whether a launch has taken them along
    // cnt_ge[N_nu + 2] (lines per centre index, for the narrow-window kernel)
    void* cnt_ws = nullptr;
    size_t cnt_ws_bytes = 0;
    size_t cnt_ge_len = 0;
    // tuning options (sdx_set_int_option)
    int64_t indexed_min_lines = 8192;  // line lists at least this long: wide lines found by centre range / the huge-line list instead of a full scan
    int64_t prepass_ticket_min_blocks = 16384;  // culled shards: from this many line blocks on the pre-pass draws its work from a counter
    int64_t mixed_precision = 0;       // 1: fp32 rational for far-wing (region I) evaluations of whole-tile windows
    int64_t segmented_raytrace = -1;   // -1: by the size of the GLOBAL grid; 0 never; 1 whenever the kernel supports the shape
    int64_t far_field = -1;            // -1: by the size of the GLOBAL grid; 0 never; 1 whenever the line kernel runs 256-point tiles
    int64_t narrow_records = -1;       // -1: by the density of the list; 1: the pre-pass writes narrow records; 0: the narrow role reads the caller's tables (long dense fp64 lists)
    // timing
    hipEvent_t t0 = nullptr, t1 = nullptr;
    void* cont_ws = nullptr;  // continuum plane [n_depth][nu_count] of the fused step
    size_t cont_ws_bytes = 0;
    // staging of the host-pointer (*_f64) entry points: one device arena and one pinned host buffer, grown on demand and kept
    void* io_dev = nullptr;
    size_t io_dev_bytes = 0;
    void* io_pin = nullptr;
    size_t io_pin_bytes = 0;
    void* xfer_pin = nullptr;  // pinned bounce buffer of sdx_memcpy_h2d / _d2h (pageable user buffers would be pinned per call)
    size_t xfer_pin_bytes = 0;
    // sdx_malloc / sdx_free: freed blocks are kept (by capacity) and handed out again — every use is ordered on this context's
    // stream, so a block can be reused the moment it is freed; a drop-in call of the Python mirror uploads ~30 small arrays
    std::mutex pool_mutex;  // guards the block pool and the bounce buffer (DeviceArray.__del__ may run on any thread)
    std::unordered_map<void*, size_t> live_blocks;  // handed out: pointer -> capacity
    std::multimap<size_t, void*> free_blocks;       // kept: capacity -> pointer
    size_t free_block_bytes = 0;
    // bumped whenever a scratch buffer is reallocated: hipGraphs captured earlier hold the old device pointers
    uint64_t ws_generation = 0;
    // two-collective mode: what sdx_synthesize_classify_dev left in the scratch (grid spacing, the shard's line ranges, the
    // continuum plane) and for which problem; the synthesis that follows with options->line_m_max checks and consumes it
    struct {
        bool valid = false;
        bool far = false;         // ... the far field was on in phase 1 and
        bool far_ranges = false;  // ... its launch computed the tiles' far ranges (far_ws)
        int n_depth = 0;
        int64_t n_nu = 0, nu_begin = 0, nu_count = 0, n_lines = 0;
        uint64_t generation = 0;
    } classified;
    bool profile = false;
    std::vector<ProfileRecord> records;
    std::vector<hipEvent_t> event_pool;
};

namespace {

constexpr size_t kSmallHeader = 4096 + 8 * (size_t)sdx::kGridSample;  // [0,2048): d_nu partials; [2048,2056): evaluation counter; [2064, ..): ticket counters; [4096, ..): grid sample

int ensure(sdx_ctx* ctx, void** buf, size_t* have, size_t need)
{
    if (*have >= need) return SDX_OK;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    hipStreamIsCapturing(ctx->stream, &st);
    if (st != hipStreamCaptureStatusNone)
        return fail(SDX_ERR_ARG, "workspace must be reserved before stream capture (sdx_reserve_line_workspace / warm-up call)");
    if (*buf) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipFree(*buf));
        *buf = nullptr;
        *have = 0;
    }
    hipError_t e = hipMalloc(buf, need);
    if (e == hipErrorOutOfMemory) {  // blocks kept by sdx_free are reclaimable: release them and try once more
        (void)hipGetLastError();
        std::lock_guard<std::mutex> lock(ctx->pool_mutex);
        if (!ctx->free_blocks.empty()) {
            hipStreamSynchronize(ctx->stream);
            for (auto& b : ctx->free_blocks) hipFree(b.second);
            ctx->free_blocks.clear();
            ctx->free_block_bytes = 0;
            e = hipMalloc(buf, need);
        }
    }
    if (e != hipSuccess) {
        *buf = nullptr;
        return fail(e == hipErrorOutOfMemory ? SDX_ERR_OOM : SDX_ERR_HIP, std::string("hipMalloc(workspace): ") + hipGetErrorString(e));
    }
    *have = need;
    ++ctx->ws_generation;
    return SDX_OK;
}

// per (line, depth) item: wide scan 16 + record 48 + slow 16 + fp32 record 32, narrow record 32, huge-line scan copy 16
size_t line_ws_need(int n_depth, int64_t n_lines) { return (size_t)n_depth * (size_t)n_lines * 160 + 256; }

LineWork carve(sdx_ctx* ctx, int n_depth, int64_t n_lines)
{
    const size_t n = (size_t)n_depth * (size_t)n_lines;
    char* p = (char*)ctx->line_ws;
    LineWork w{};
    w.wrec = (WideRec*)p;                    // 48 n, 16-byte aligned
    w.wscan = (WideScan*)(w.wrec + n);       // 16 n
    w.wslow = (WideSlow*)(w.wscan + n);      // 16 n
    WideRec32* rec32 = (WideRec32*)(w.wslow + n);  // 32 n
    w.wrec32 = ctx->mixed_precision ? rec32 : nullptr;
    w.n_inv = (double*)(rec32 + n);  // 8 n each
    w.n_y = w.n_inv + n;
    w.n_amp = w.n_y + n;
    w.nhw = (unsigned char*)(w.n_amp + n);  // 1 n (the 8 n bytes of the former window bounds stay reserved)
    w.skip_unlisted_scan = 0;
    w.n_inv32 = w.n_y32 = w.n_amp32 = nullptr;
    w.lnu32 = nullptr;
    if (ctx->mixed_precision) {  // the narrow role's records as floats in the same 24 n bytes, + the line frequencies as float pairs
        w.n_inv32 = (float*)w.n_inv;
        w.n_y32 = w.n_inv32 + n;
        w.n_amp32 = w.n_y32 + n;
        w.lnu32 = (float2v*)(w.n_amp32 + n);
    }
    w.hscan = (WideScan*)((int*)(w.n_amp + n) + 2 * n);  // 16 n (only the first hcount[0] entries of a row are used)
    w.cnt_ge = (int*)ctx->cnt_ws;
    w.centre = w.cnt_ge + ctx->cnt_ge_len;
    w.nhw_max = w.centre + n_lines;
    w.whw_max = w.nhw_max + n_lines;
    w.hlist = nullptr;  // set for long lists (set_line_lists)
    w.wlist = nullptr;
    w.wrank = nullptr;
    w.xlist = nullptr;
    w.hcount = w.whw_max + 5 * n_lines + 8;  // behind hlist [n], wlist [n], wrank [n + 1], xlist [n]
    w.evals = (unsigned long long*)((char*)ctx->small_ws + 2048);
    return w;
}

hipEvent_t take_event(sdx_ctx* ctx)
{
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

struct LaunchScope {
    sdx_ctx* ctx;
    ProfileRecord rec{};
    bool on;
    LaunchScope(sdx_ctx* c, const char* name, const char* variant = nullptr) : ctx(c), on(c->profile)
    {
        if (on) {
            rec.name = name;
            rec.variant = variant;
            rec.start = take_event(ctx);
            rec.stop = take_event(ctx);
            hipEventRecord(rec.start, ctx->stream);
        }
    }
    ~LaunchScope()
    {
        if (on) {
            hipEventRecord(rec.stop, ctx->stream);
            ctx->records.push_back(rec);
        }
    }
};

int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SDX_ERR_HIP, std::string(what) + " launch: " + hipGetErrorString(e));
    return SDX_OK;
}

inline dim3 grid2(int64_t n, int rows) { return dim3((unsigned)((n + kBlock - 1) / kBlock), (unsigned)rows); }
inline unsigned blocks1(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

// d_nu partial maxima into small_ws
int launch_dnu(sdx_ctx* ctx, int64_t n_nu, const double* nus, int* n_partial, int* zero = nullptr, int64_t n_zero = 0)
{
    int rc = ensure(ctx, &ctx->small_ws, &ctx->small_ws_bytes, kSmallHeader);
    if (rc) return rc;
    ctx->classified.valid = false;  // (the grid-spacing partials of a two-collective phase 1 are overwritten)
    const int nb = (int)std::min<int64_t>(kDnuPartials, std::max<int64_t>(1, (n_nu + kBlock * 8 - 1) / (kBlock * 8)));
    {
        LaunchScope ls(ctx, "k_dnu_partial");
        hipLaunchKernelGGL(k_dnu_partial, dim3(nb), dim3(kBlock), 0, ctx->stream, n_nu, nus, (double*)ctx->small_ws, zero, n_zero, ctx->far_req);
        if (ctx->far_req.count) ctx->far_req_done = true;
    }
    *n_partial = nb;
    return check_launch("k_dnu_partial");
}

static int check_file_planes(const sdx_continuum* c, int64_t n_nu)
{
    REQUIRE(c->n_file_planes >= 0 && c->n_file_planes <= 4, "continuum: n_file_planes must be 0..4");
    for (int k = 0; k < c->n_file_planes; ++k) REQUIRE(c->file_plane[k], "continuum: null file plane");
    REQUIRE(c->n_file_planes == 0 || c->file_plane_ld >= n_nu, "continuum: file_plane_ld must cover the grid");
    return SDX_OK;
}

ContinuumArgs to_args(const sdx_continuum* c, const double* bf_coef)
{
    ContinuumArgs a{};
    a.lambdas = c->lambdas;
    a.n_table = c->n_table;
    a.table_wavelength = c->table_wavelength;
    a.table_sigma = c->table_sigma;
    a.table_density = c->table_density;
    a.bf_n_species = c->bf_cutoff ? c->bf_n_species : 0;
    a.bf_species_offsets = c->bf_species_offsets;
    a.bf_species_ion_number = c->bf_species_ion_number;
    a.bf_cutoff = c->bf_cutoff;
    a.bf_n_levels = a.bf_n_species > 0 ? c->bf_n_levels : 0;
    a.bf_coef = bf_coef;
    a.ff_n_species = c->ff_number_density ? c->ff_n_species : 0;
    a.ff_species_ion_number = c->ff_species_ion_number;
    a.ff_number_density = c->ff_number_density;
    a.ray_n_h = c->ray_n_h;
    a.ray_n_he = c->ray_n_he;
    a.ray_n_h2 = c->ray_n_h2;
    a.rayleigh_enabled = c->rayleigh_enabled;
    a.electron_density = c->electron_density;
    a.temperature = c->temperature;
    a.n_file_planes = std::max(0, std::min(4, c->n_file_planes));
    for (int k = 0; k < a.n_file_planes; ++k) a.file_plane[k] = c->file_plane[k];
    a.file_plane_ld = c->file_plane_ld;
    return a;
}

// bf coefficients [n_levels][n_depth] into small_ws after the header; n_levels must be known on the host
int launch_bf_coef(sdx_ctx* ctx, int n_depth, int n_species, int n_levels, const int32_t* offs, const int32_t* ions,
                   const double* cutoff, const double* level_density, double** coef_out)
{
    const size_t need = kSmallHeader + (size_t)n_levels * n_depth * sizeof(double);
    int rc = ensure(ctx, &ctx->small_ws, &ctx->small_ws_bytes, need);
    if (rc) return rc;
    double* coef = (double*)((char*)ctx->small_ws + kSmallHeader);
    {
        LaunchScope ls(ctx, "k_bf_coef");
        hipLaunchKernelGGL(k_bf_coef, dim3(blocks1((int64_t)n_levels * n_depth)), dim3(kBlock), 0, ctx->stream, n_depth,
                           n_species, offs, ions, cutoff, level_density, coef);
    }
    *coef_out = coef;
    return check_launch("k_bf_coef");
}

}  // namespace

// ================================================================================================ runtime
extern "C" {

const char* sdx_version(void) { return "stardis_hip 0.1 (gfx950, fp64)"; }
const char* sdx_last_error_string(void) { return g_error.c_str(); }
int sdx_last_error_code(void) { return g_error_code; }

int sdx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int sdx_set_device(int device)
{
    HIP_TRY(hipSetDevice(device));
    return SDX_OK;
}

sdx_ctx* sdx_create(int device, void* stream)
{
    if (hipSetDevice(device) != hipSuccess) {
        fail(SDX_ERR_HIP, "hipSetDevice failed: no usable HIP device " + std::to_string(device));
        (void)hipGetLastError();
        return nullptr;
    }
    sdx_ctx* ctx = new sdx_ctx();
    ctx->device = device;
    if (hipDeviceGetAttribute(&ctx->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ctx->n_cu <= 0) ctx->n_cu = 256;
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            fail(SDX_ERR_HIP, "hipStreamCreate failed");
            delete ctx;
            return nullptr;
        }
        ctx->own_stream = true;
    }
    hipEventCreate(&ctx->t0);
    hipEventCreate(&ctx->t1);
    return ctx;
}

void sdx_destroy(sdx_ctx* ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (auto& r : ctx->records) {
        hipEventDestroy(r.start);
        hipEventDestroy(r.stop);
    }
    for (auto e : ctx->event_pool) hipEventDestroy(e);
    if (ctx->cont_ws) hipFree(ctx->cont_ws);
    if (ctx->t0) hipEventDestroy(ctx->t0);
    if (ctx->t1) hipEventDestroy(ctx->t1);
    if (ctx->line_ws) hipFree(ctx->line_ws);
    if (ctx->small_ws) hipFree(ctx->small_ws);
    if (ctx->part_ws) hipFree(ctx->part_ws);
    if (ctx->far_ws) hipFree(ctx->far_ws);
    if (ctx->cnt_ws) hipFree(ctx->cnt_ws);
    if (ctx->io_dev) hipFree(ctx->io_dev);
    if (ctx->io_pin) hipHostFree(ctx->io_pin);
    if (ctx->xfer_pin) hipHostFree(ctx->xfer_pin);
    for (auto& b : ctx->free_blocks) hipFree(b.second);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

int sdx_set_stream(sdx_ctx* ctx, void* stream)
{
    REQUIRE(ctx, "null context");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) {
        hipStreamDestroy(ctx->stream);
        ctx->own_stream = false;
    }
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    return SDX_OK;
}

void* sdx_get_stream(sdx_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int sdx_set_int_option(sdx_ctx* ctx, const char* name, int64_t value)
{
    REQUIRE(ctx && name, "sdx_set_int_option: null pointer");
    if (std::strcmp(name, "indexed_min_lines") == 0) {
        ctx->indexed_min_lines = value;
        return SDX_OK;
    }
    if (std::strcmp(name, "prepass_ticket_min_blocks") == 0) {
        ctx->prepass_ticket_min_blocks = value;
        return SDX_OK;
    }
    if (std::strcmp(name, "mixed_precision") == 0) {
        ctx->mixed_precision = value ? 1 : 0;
        return SDX_OK;
    }
    if (std::strcmp(name, "segmented_raytrace") == 0) {
        ctx->segmented_raytrace = value < 0 ? -1 : (value ? 1 : 0);
        return SDX_OK;
    }
    if (std::strcmp(name, "far_field") == 0) {
        ctx->far_field = value < 0 ? -1 : (value ? 1 : 0);
        return SDX_OK;
    }
    if (std::strcmp(name, "narrow_records") == 0) {
        ctx->narrow_records = value < 0 ? -1 : (value ? 1 : 0);
        return SDX_OK;
    }
    return fail(SDX_ERR_ARG, std::string("unknown option ") + name);
}

static bool far_field_on(const sdx_ctx* ctx, int64_t n_nu_global);
int sdx_far_field_active(const sdx_ctx* ctx, int64_t n_nu_global)
{
    REQUIRE(ctx, "null context");
    return far_field_on(ctx, n_nu_global) ? 1 : 0;
}

int sdx_far_field_rule(int64_t* min_points, int* tile_points, int* near_points)
{
    if (min_points) *min_points = 32768;  // kFarMinPoints (static_assert where it is defined)
    if (tile_points) *tile_points = kFarTile;
    // a (line, tile) pair is far when the line's centre is >= kFarRatio tile half-widths from the tile's centre: a point closer than
    // that + half a tile to the centre is always evaluated where it lies
    if (near_points) *near_points = (int)(kFarRatio * (kFarTile / 2)) + kFarTile / 2;
    return SDX_OK;
}

int sdx_synchronize(sdx_ctx* ctx)
{
    REQUIRE(ctx, "null context");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SDX_OK;
}

constexpr size_t kBlockPoolLimit = (size_t)4 << 30;  // bytes of freed blocks kept per context
static size_t block_capacity(size_t bytes)
{
    if (bytes <= 256) return 256;
    if (bytes >= ((size_t)1 << 20)) return (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);  // whole MiB
    size_t c = 256;
    while (c < bytes) c <<= 1;
    return c;
}

void* sdx_malloc(sdx_ctx* ctx, size_t bytes)
{
    if (!ctx) return nullptr;
    hipSetDevice(ctx->device);
    const size_t cap = block_capacity(bytes);
    std::lock_guard<std::mutex> lock(ctx->pool_mutex);
    auto it = ctx->free_blocks.lower_bound(cap);
    if (it != ctx->free_blocks.end() && it->first <= 2 * cap) {  // a kept block of (nearly) this size
        void* p = it->second;
        ctx->free_block_bytes -= it->first;
        ctx->live_blocks[p] = it->first;
        ctx->free_blocks.erase(it);
        return p;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, cap);
    if (e != hipSuccess && !ctx->free_blocks.empty()) {  // out of memory with blocks in the pool: release them and try again
        hipStreamSynchronize(ctx->stream);
        for (auto& b : ctx->free_blocks) hipFree(b.second);
        ctx->free_blocks.clear();
        ctx->free_block_bytes = 0;
        e = hipMalloc(&p, cap);
    }
    if (e != hipSuccess) {
        fail(SDX_ERR_OOM, std::string("hipMalloc: ") + hipGetErrorString(e));
        return nullptr;
    }
    ctx->live_blocks[p] = cap;
    return p;
}

int sdx_free(sdx_ctx* ctx, void* ptr)
{
    REQUIRE(ctx, "null context");
    if (!ptr) return SDX_OK;
    {
        std::lock_guard<std::mutex> lock(ctx->pool_mutex);
        auto it = ctx->live_blocks.find(ptr);
        if (it == ctx->live_blocks.end()) {
            // not handed out by sdx_malloc of this context (or freed already: the block may be pooled or reused by now)
            return fail(SDX_ERR_ARG, "sdx_free: pointer is not a live block of this context (double free?)");
        }
        const size_t cap = it->second;
        ctx->live_blocks.erase(it);
        if (ctx->free_block_bytes + cap <= kBlockPoolLimit) {
            ctx->free_blocks.emplace(cap, ptr);
            ctx->free_block_bytes += cap;
            return SDX_OK;
        }
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(ptr));
    return SDX_OK;
}

// Copies between pageable user memory and the device go through a pinned bounce buffer of the context, in chunks: handing the
// runtime a pageable pointer makes it pin the user's pages for the call — measured 9 ms for a 61 KB numpy array, where the
// bounce costs a memcpy and a 10 us DMA.
constexpr size_t kXferChunk = (size_t)8 << 20;
static int xfer_buffer(sdx_ctx* ctx, size_t bytes)
{
    const size_t want = std::min(kXferChunk, std::max(bytes, (size_t)1 << 20));
    if (ctx->xfer_pin_bytes >= want) return SDX_OK;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->xfer_pin) HIP_TRY(hipHostFree(ctx->xfer_pin));
    ctx->xfer_pin = nullptr;
    ctx->xfer_pin_bytes = 0;
    HIP_TRY(hipHostMalloc(&ctx->xfer_pin, want, hipHostMallocDefault));
    ctx->xfer_pin_bytes = want;
    return SDX_OK;
}

int sdx_memcpy_h2d(sdx_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    REQUIRE(ctx && (bytes == 0 || (dst && src)), "sdx_memcpy_h2d: null pointer");
    if (bytes == 0) return SDX_OK;
    static const bool no_pin = knob("SDX_NO_PINNED_STAGING") != nullptr;
    if (no_pin) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));  // pageable source: safe to reuse on return
        return SDX_OK;
    }
    std::lock_guard<std::mutex> lock(ctx->pool_mutex);  // one bounce buffer per context
    int rc = xfer_buffer(ctx, bytes);
    if (rc) return rc;
    for (size_t off = 0; off < bytes; off += ctx->xfer_pin_bytes) {
        const size_t n = std::min(ctx->xfer_pin_bytes, bytes - off);
        std::memcpy(ctx->xfer_pin, (const char*)src + off, n);
        HIP_TRY(hipMemcpyAsync((char*)dst + off, ctx->xfer_pin, n, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));  // the bounce buffer is reused by the next chunk / call
    }
    return SDX_OK;
}

int sdx_memcpy_d2h(sdx_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    REQUIRE(ctx && (bytes == 0 || (dst && src)), "sdx_memcpy_d2h: null pointer");
    if (bytes == 0) return SDX_OK;
    static const bool no_pin = knob("SDX_NO_PINNED_STAGING") != nullptr;
    if (no_pin) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return SDX_OK;
    }
    std::lock_guard<std::mutex> lock(ctx->pool_mutex);
    int rc = xfer_buffer(ctx, bytes);
    if (rc) return rc;
    for (size_t off = 0; off < bytes; off += ctx->xfer_pin_bytes) {
        const size_t n = std::min(ctx->xfer_pin_bytes, bytes - off);
        HIP_TRY(hipMemcpyAsync(ctx->xfer_pin, (const char*)src + off, n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        std::memcpy((char*)dst + off, ctx->xfer_pin, n);
    }
    return SDX_OK;
}

// Page-locked host memory for callers that keep their own staging / result buffers there (the Python mirror's fused call): the
// copies below move it by DMA directly, without the bounce buffer and its memcpy.
void* sdx_host_alloc(sdx_ctx* ctx, size_t bytes)
{
    if (!ctx || bytes == 0) {
        fail(SDX_ERR_ARG, "sdx_host_alloc: null context or zero size");
        return nullptr;
    }
    void* p = nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        fail(SDX_ERR_OOM, "sdx_host_alloc: hipHostMalloc of " + std::to_string(bytes) + " bytes failed");
        return nullptr;
    }
    return p;
}

int sdx_host_free(void* ptr)
{
    if (!ptr) return SDX_OK;
    HIP_TRY(hipHostFree(ptr));
    return SDX_OK;
}

int sdx_memcpy_h2d_pinned(sdx_ctx* ctx, void* dst, const void* src_pinned, size_t bytes)
{
    REQUIRE(ctx && (bytes == 0 || (dst && src_pinned)), "sdx_memcpy_h2d_pinned: null pointer");
    if (bytes == 0) return SDX_OK;
    HIP_TRY(hipMemcpyAsync(dst, src_pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
    return SDX_OK;
}

int sdx_memcpy_d2h_pinned(sdx_ctx* ctx, void* dst_pinned, const void* src, size_t bytes)
{
    REQUIRE(ctx && (bytes == 0 || (dst_pinned && src)), "sdx_memcpy_d2h_pinned: null pointer");
    if (bytes == 0) return SDX_OK;
    HIP_TRY(hipMemcpyAsync(dst_pinned, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SDX_OK;
}

int sdx_memset(sdx_ctx* ctx, void* dst, int value, size_t bytes)
{
    REQUIRE(ctx && (bytes == 0 || dst), "sdx_memset: null pointer");
    if (bytes == 0) return SDX_OK;
    HIP_TRY(hipMemsetAsync(dst, value, bytes, ctx->stream));
    return SDX_OK;
}

int sdx_reserve_line_workspace(sdx_ctx* ctx, int n_depth, int64_t n_lines)
{
    REQUIRE(ctx && n_depth > 0 && n_lines >= 0, "sdx_reserve_line_workspace: bad sizes");
    int rc = ensure(ctx, &ctx->small_ws, &ctx->small_ws_bytes, kSmallHeader);
    if (rc) return rc;
    return ensure(ctx, &ctx->line_ws, &ctx->line_ws_bytes, line_ws_need(n_depth, n_lines));
}

int sdx_graph_begin(sdx_ctx* ctx)
{
    REQUIRE(ctx, "null context");
    HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    return SDX_OK;
}

// A captured graph has the context's scratch pointers baked in: the handle remembers the workspace generation it was
// captured under and sdx_graph_launch refuses to replay it after any scratch buffer has been reallocated since.
struct GraphHandle {
    hipGraphExec_t exec;
    uint64_t generation;
};

int sdx_graph_end(sdx_ctx* ctx, void** graph_exec_out)
{
    REQUIRE(ctx && graph_exec_out, "sdx_graph_end: null pointer");
    hipGraph_t graph = nullptr;
    HIP_TRY(hipStreamEndCapture(ctx->stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(SDX_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    *graph_exec_out = (void*)new GraphHandle{exec, ctx->ws_generation};
    return SDX_OK;
}

int sdx_graph_launch(sdx_ctx* ctx, void* graph_exec)
{
    REQUIRE(ctx && graph_exec, "sdx_graph_launch: null pointer");
    GraphHandle* h = (GraphHandle*)graph_exec;
    if (h->generation != ctx->ws_generation)
        return fail(SDX_ERR_STALE, "graph is stale: the context's workspace was reallocated after the capture (a larger problem ran on the "
                                   "same context); capture again");
    HIP_TRY(hipGraphLaunch(h->exec, ctx->stream));
    return SDX_OK;
}

int sdx_graph_destroy(sdx_ctx* ctx, void* graph_exec)
{
    REQUIRE(ctx, "null context");
    if (graph_exec) {
        GraphHandle* h = (GraphHandle*)graph_exec;
        hipError_t e = hipGraphExecDestroy(h->exec);
        delete h;
        if (e != hipSuccess) return fail(SDX_ERR_HIP, std::string("hipGraphExecDestroy: ") + hipGetErrorString(e));
    }
    return SDX_OK;
}

int sdx_timer_start(sdx_ctx* ctx)
{
    REQUIRE(ctx, "null context");
    HIP_TRY(hipEventRecord(ctx->t0, ctx->stream));
    return SDX_OK;
}

int sdx_timer_stop(sdx_ctx* ctx, double* elapsed_ms)
{
    REQUIRE(ctx && elapsed_ms, "sdx_timer_stop: null pointer");
    HIP_TRY(hipEventRecord(ctx->t1, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->t1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->t0, ctx->t1));
    *elapsed_ms = ms;
    return SDX_OK;
}

int sdx_profile_enable(sdx_ctx* ctx, int on)
{
    REQUIRE(ctx, "null context");
    ctx->profile = on != 0;
    return SDX_OK;
}

int sdx_profile_reset(sdx_ctx* ctx)
{
    REQUIRE(ctx, "null context");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (auto& r : ctx->records) {
        ctx->event_pool.push_back(r.start);
        ctx->event_pool.push_back(r.stop);
    }
    ctx->records.clear();
    return SDX_OK;
}

int sdx_profile_get(sdx_ctx* ctx, const char* kernel, int64_t* launches, double* total_ms)
{
    REQUIRE(ctx && kernel && launches && total_ms, "sdx_profile_get: null pointer");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    int64_t n = 0;
    double tot = 0.0;
    for (auto& r : ctx->records) {
        if (std::strcmp(r.name, kernel) != 0) continue;
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, r.start, r.stop));
        tot += ms;
        ++n;
    }
    *launches = n;
    *total_ms = tot;
    return SDX_OK;
}

int sdx_profile_variant(sdx_ctx* ctx, const char* kernel, char* out, int out_len)
{
    REQUIRE(ctx && kernel && out && out_len > 0, "sdx_profile_variant: null pointer");
    out[0] = 0;
    for (auto& r : ctx->records)
        if (std::strcmp(r.name, kernel) == 0 && r.variant) std::snprintf(out, (size_t)out_len, "%s", r.variant);
    return SDX_OK;
}

// ================================================================================================ line opacity
struct ContinuumJob {  // continuum plane computed by the trailing blocks of the pre-pass launch
    const sdx_continuum* cont;
    int64_t nu_begin, nu_count;
    double* plane;
};
// Frequency shards of long lists, two-collective mode: the classification stream of a culled pre-pass (the largest (gamma + dw) alpha
// of every line, 8 N_l (2 N_d + G) bytes read by EVERY rank) split over the ranks — phase 1 classifies a share of the lines into the
// caller's m_max array (and does everything else of that launch: grid spacing, the shard's line ranges, the continuum plane), the
// caller exchanges the shares (an all-gather of 8 N_l bytes), phase 2 is the rest of the step on the gathered array.
struct ClassifyPhase {
    int phase;             // 1 or 2
    int64_t begin, count;  // phase 1: the lines to classify
    double* m_max;         // [n_lines]: phase 1 writes [begin, begin + count), phase 2 reads everything
};

// The segmented formal solution (k_raytrace_seg: the gaps of a ray over the 8 waves of a workgroup) pays ~40 % more
// instructions for eight times the waves: it wins where k_raytrace would leave the chip under three waves per SIMD.
static const int kSegWaves = knob("SDX_RT_NS") && std::atoi(knob("SDX_RT_NS")) == 4 ? 4 : 8;  // experiment knob: 4 waves x 14 gaps
static const int kSegMax = kSegWaves == 4 ? 14 : 7;
static size_t seg_lds_doubles(int n_depth, int nth)
{
    const int gpw = 64 / nth;
    // segment maps, then the larger of the staging arrays (transposed ray table with an odd row stride, then — 16-byte aligned — the
    // (source, sqrt(alpha)) pairs) and the flux terms that reuse their space
    return (size_t)kSegWaves * 128 + std::max((((size_t)nth * ((n_depth - 1) | 1) + 1) & ~(size_t)1) + (size_t)2 * gpw * n_depth, (size_t)kSegWaves * kSegMax * gpw * nth);
}
// Which kernel runs must not depend on how the grid is sharded or on the device (the two differ by the rounding of the affine
// composition, a few ulp: a shard below the threshold next to an unsharded run above it would break the bit-identity of
// sharded and unsharded spectra): the choice is made from the GLOBAL grid size against a fixed constant — three k_raytrace
// waves per SIMD of a 256-CU part — or set explicitly (context option "segmented_raytrace").
constexpr int64_t kSegLegacyWaves = (int64_t)3 * 4 * 256;
static bool use_segmented_raytrace(const sdx_ctx* ctx, int n_depth, int64_t n_nu_global, int n_theta, bool plain)
{
    static const int env_mode = knob("SDX_RT_SEG") ? std::atoi(knob("SDX_RT_SEG")) : -1;  // A/B knob: 0 never, 1 whenever possible
    const int mode = ctx->segmented_raytrace >= 0 ? (int)ctx->segmented_raytrace : env_mode;
    if (mode == 0 || !plain || n_theta > 64) return false;
    if ((n_depth - 1 + kSegWaves - 1) / kSegWaves > kSegMax || seg_lds_doubles(n_depth, n_theta) * sizeof(double) > 64 * 1024) return false;
    if (mode == 1) return true;
    const int64_t legacy_waves = (n_nu_global + 64 / n_theta - 1) / (64 / n_theta);
    return legacy_waves < kSegLegacyWaves;
}

// long lists: hlist / wlist / wrank from whw_max (two small launches)
static void launch_line_lists(sdx_ctx* ctx, int64_t n_lines, LineWork& w, int pre_lines, const ClassSource* from_classification = nullptr)
{
    ClassSource cs{};
    if (from_classification) cs = *from_classification;
    else cs.whw_max = w.whw_max;
    w.hlist = w.whw_max + n_lines;
    w.wlist = w.hlist + n_lines;
    w.wrank = w.wlist + n_lines;
    w.xlist = w.sel ? w.wrank + n_lines + 1 : nullptr;  // culled runs only
    const unsigned hb = (unsigned)((n_lines + kHlistBlock - 1) / kHlistBlock);
    int* block_cnt = w.hcount + 16;  // 3 hb ints behind the counters and sel (reserved in cnt_ws)
    hipLaunchKernelGGL(k_hlist_count, dim3(hb), dim3(kHlistBlock), 0, ctx->stream, n_lines, cs, block_cnt, w.sel, pre_lines, w.ticket);
    hipLaunchKernelGGL(k_hlist_scatter, dim3(hb), dim3(kHlistBlock), 0, ctx->stream, n_lines, cs, (const int*)block_cnt, w.hlist,
                       w.wlist, w.wrank, w.hcount, w.xlist, w.sel, pre_lines);
}

static int line_prepass(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t n_lines, const double* line_nus,
                        const double* doppler, const double* gammas, int gamma_cols, const double* alphas, bool fill_work,
                        int32_t* lo_ref, int32_t* hi_ref, LineWork* w_out, bool count_evals = true,
                        const ContinuumJob* job = nullptr, const LineParams* gen = nullptr, int64_t nu_begin = 0, int64_t nu_count = -1,
                        const ClassifyPhase* ph = nullptr)
{
    if (nu_count < 0) nu_count = n_nu;
    const LineParams lp = gen ? *gen : LineParams{};
    int n_partial = 0;
    // every pre-pass rewrites the scratch a two-collective phase 1 leaves for its phase 2 (grid spacing, line ranges, lists): only the
    // phase 2 that consumes it may find it valid (phase 1 sets the flag again when its launch is enqueued)
    const bool consumes_classified = ph && ph->phase == 2;
    if (!consumes_classified) ctx->classified.valid = false;
    int rc = ensure(ctx, &ctx->small_ws, &ctx->small_ws_bytes, kSmallHeader);
    if (rc) return rc;
    const bool scan_in_block = n_nu <= 16384;  // every pre-pass block re-scans a small grid instead of a separate launch
    LineWork w{};
    // 16 lines per block while all such blocks are resident at once (two 1024-thread blocks per CU), else 32
    int pre_lines = ((n_lines + 15) / 16) * ((n_depth + kPreDepths - 1) / kPreDepths) <= 2 * (int64_t)ctx->n_cu ? 16 : 32;
    // A culled shard (decided below from the same quantities) prepares its own line range + the listed lines: at an eighth of 1.5e5
    // lines ~22 000 lines, 690 blocks of 32 on the chip's 512 slots of 1024 threads — a whole second round of the block's chain of
    // round trips for a third of a round's work.  48 lines per block (three items per thread, registers only) are 470 blocks:
    // ONE round, a chain one item longer.
    static const bool no_cull_early = knob("SDX_NO_CULL") != nullptr;
    static const int pre_lines_env = knob("SDX_PRE_LINES") ? std::atoi(knob("SDX_PRE_LINES")) : 0;  // experiment knob: 32, 48 or 64 on culled shards
    const bool will_cull = fill_work && !no_cull_early && n_lines >= ctx->indexed_min_lines && !count_evals && !gen && n_nu > 16384 && nu_count < n_nu;
    // (lists long enough for the counter-driven launch keep 32: its looping kernel has no registers to spare for a third item)
    // (54 when the model has at most 56 depth points — every MARCS model — and three items per thread still cover the block)
    if (will_cull && !lo_ref && (n_lines + 31) / 32 < ctx->prepass_ticket_min_blocks)
        pre_lines = (pre_lines_env == 32 || pre_lines_env == 48) ? pre_lines_env : (n_depth <= 56 ? 54 : 48);
    const int n_line_blocks = (int)((n_lines + pre_lines - 1) / pre_lines);
    int n_pixel_blocks = 0;
    if (fill_work) {
        rc = ensure(ctx, &ctx->line_ws, &ctx->line_ws_bytes, line_ws_need(n_depth, n_lines));
        if (rc) return rc;
        ctx->cnt_ge_len = (size_t)(n_nu + 2 + 63) / 64 * 64;
        // cnt_ge, then per line: centre, nhw_max, whw_max, hlist, wlist, wrank, xlist; then hcount, sel and the per-block counts
        // ... and, behind those, the classification pass's per-line maxima (doubles)
        rc = ensure(ctx, &ctx->cnt_ws, &ctx->cnt_ws_bytes, (ctx->cnt_ge_len + 10 * (size_t)n_lines + 128 + 3 * ((size_t)n_lines / 1024 + 8) + 8) * sizeof(int));
        if (rc) return rc;
        w = carve(ctx, n_depth, n_lines);
        if (n_depth > kPreDepths) HIP_TRY(hipMemsetAsync(w.nhw_max, 0, (size_t)2 * n_lines * sizeof(int), ctx->stream));  // nhw_max and whw_max
        if (count_evals) HIP_TRY(hipMemsetAsync(w.evals, 0, sizeof(unsigned long long), ctx->stream));
        else w.evals = nullptr;
        n_pixel_blocks = (int)((n_nu + 2 + kPreBlock - 1) / kPreBlock);
    }
    w.sel = nullptr;
    w.gather = 0;
    w.ticket = nullptr;
    w.front = 0;
    // VERY dense long fp64 lists (>= 4 lines per grid point: the rule of the narrow role's subsets): no narrow records — the narrow role
    // reads the caller's three tables itself (LineWork::narrow_raw).  Pure scheduling (the same three operations form 1 / dw, y and the
    // amplitude either way).  Measured in round 6 (profiles/r06_raw.txt): 1e6 lines on 120 398 points — pre-pass 904 -> 721 us (a stream:
    // 2.13 GB of traffic counted, 1.31 GB of it these records), line kernel 5.93 -> 6.07 ms (ten more instructions per evaluated line), step
    // 7.27 -> 7.24 ms and 1.35 GB less scratch; 1.5e5 lines: pre-pass 165 -> 149 us, line kernel 1.218 -> 1.241 ms, the step 7 us SLOWER —
    // hence the density in the rule.
    static const int narrow_records_env = knob("SDX_NARROW_RECORDS") ? std::atoi(knob("SDX_NARROW_RECORDS")) : -1;  // A/B knob; the option "narrow_records" wins
    const int narrow_records = ctx->narrow_records >= 0 ? (int)ctx->narrow_records : narrow_records_env;
    const bool very_dense = 2 * n_lines >= 8 * n_nu;
    if (fill_work && !gen && !ctx->mixed_precision && n_lines >= ctx->indexed_min_lines && (narrow_records == 0 || (very_dense && narrow_records != 1))) {
        w.narrow_raw = gamma_cols > 1 ? n_depth : 1;
        w.n_inv = const_cast<double*>(doppler);
        w.n_y = const_cast<double*>(gammas);
        w.n_amp = const_cast<double*>(alphas);
    }
    // long lists find their wide lines through hlist / wlist: the scan words of a line without a wide window anywhere are never
    // read (needs the line's widest window inside one block: one depth block per line)
    w.skip_unlisted_scan = (fill_work && n_lines >= ctx->indexed_min_lines && n_depth <= kPreDepths) ? 1 : 0;
    w.n_pix = n_pixel_blocks;
    w.shard_begin = nu_begin;
    w.shard_end = nu_begin + nu_count;
    // A frequency shard of a long list does not need every line prepared: a streaming classification pass finds the lines
    // that can reach any column (-> hlist), the full pre-pass then runs on the lines centred near the shard plus those.
    // Everything is decided on the device (no host round trip, graph-capturable); which lines a shard prepares does not
    // change what it computes for them.
    static const bool no_cull = knob("SDX_NO_CULL") != nullptr;
    const bool cull = fill_work && !no_cull && n_lines >= ctx->indexed_min_lines && !count_evals && !gen && !scan_in_block && nu_count < n_nu;
    REQUIRE(!ph || cull, "two-collective mode is for frequency shards of long dense line lists (>= indexed_min_lines lines, a grid of more than "
                         "16384 points, nu_count < n_nu, no evaluation count)");
    // the grid-spacing reduction: a launch of its own, or — culled runs — the first blocks of the classification launch
    int* const sel = cull ? w.hcount + 4 : nullptr;
    // (grid-spacing blocks of a classification launch: four pairs of loads per thread — one trip of their loop — up to 256 blocks)
    if (cull) n_partial = (int)std::min<int64_t>(kDnuPartials, std::max<int64_t>(1, (n_nu + kBlock * 4 - 1) / (kBlock * 4)));
    else if (!scan_in_block && (rc = launch_dnu(ctx, n_nu, nus, &n_partial))) return rc;
    // continuum blocks of the fused step: one per (frequency tile of `threads` points, group of dgs depths) when the per-depth
    // factors of a group fit LDS (always, for a handful of bound-free levels), else one per (tile, depth) evaluating every point
    // from scratch
    struct ContPlan {
        ContinuumArgs ca;
        size_t shmem;
        int cont_tiles, stage_table;
        unsigned cont_rows;
        bool tiled;
    };
    auto plan_continuum = [&](int threads, ContPlan* p) -> int {
        p->ca = to_args(job->cont, nullptr);
        ContinuumArgs& ca = p->ca;
        ca.bf_level_density = job->cont->bf_level_density;
        p->shmem = 8;
        if (ca.bf_n_species > 0) {
            REQUIRE(job->cont->bf_n_levels > 0 && job->cont->bf_n_levels <= 4096, "synthesize: bf_n_levels must be set (1..4096)");
            p->shmem = (size_t)job->cont->bf_n_levels * sizeof(double);
        }
        p->stage_table = ca.table_sigma && ca.n_table > 0 && ca.n_table <= 1024;  // 1-D cross-section table searched from LDS
        if (p->stage_table) p->shmem = ((ca.bf_n_species > 0 ? (size_t)job->cont->bf_n_levels : 0) + 2 * (size_t)ca.n_table) * sizeof(double);
        p->cont_tiles = (int)((job->nu_count + (int64_t)threads * kContPoints - 1) / ((int64_t)threads * kContPoints));
        const size_t n_lev = ca.bf_n_species > 0 ? (size_t)job->cont->bf_n_levels : 0;
        const size_t tile_shmem = ((size_t)kContDepths * (n_lev + 6) + (p->stage_table ? 2 * (size_t)ca.n_table : 0)) * sizeof(double);
        p->cont_rows = (unsigned)n_depth;
        p->tiled = false;
        static const int cont_dgs_env = knob("SDX_CONT_DGS") ? std::atoi(knob("SDX_CONT_DGS")) : -1;  // experiment knob: 0 = per-point blocks
        if (tile_shmem <= 48 * 1024 && cont_dgs_env != 0) {
            // depths per block: as many as leave ~2 x 1024 threads of such blocks per CU (one depth per block on small grids)
            int dgs = (int)std::max<int64_t>(1, std::min<int64_t>(kContDepths, ((int64_t)p->cont_tiles * threads / kPreBlock * n_depth) / (2 * (int64_t)ctx->n_cu)));
            if (threads == kPreBlock) {
                // ... next to the launch's pre-pass blocks: ALL blocks of the launch in one round of the chip's 2 x n_cu slots of 1024
                // threads (S-c2: 133 pre-pass blocks + 448 one-depth tiles were 581 on 512 slots — the 69 of the second round ended
                // the launch at 15.6 us where two depths per tile end it at 12.0; device time stamps, scripts/r5/pre_stats.sh)
                const int64_t slots = std::max<int64_t>(1, 2 * (int64_t)ctx->n_cu - ((int64_t)n_line_blocks + n_pixel_blocks) * ((n_depth + kPreDepths - 1) / kPreDepths));
                const int64_t fit = ((int64_t)p->cont_tiles * n_depth + slots - 1) / slots;
                dgs = (int)std::max<int64_t>(dgs, std::min<int64_t>(kContDepths, fit));
            }
            // riding with the classification stream (256-thread blocks) the chip is full anyway: as many depths per block as
            // there are — the per-frequency work (table search, nu^-3, Rayleigh powers) is then shared by eight points
            if (threads == kBlock) dgs = kContDepths;
            if (cont_dgs_env > 0) dgs = std::min(cont_dgs_env, kContDepths);
            p->stage_table |= 2 | (dgs << 4);
            p->shmem = tile_shmem;
            p->cont_rows = (unsigned)((n_depth + dgs - 1) / dgs);
            p->tiled = true;
        }
        return SDX_OK;
    };
    // culled runs: classification stream, the line lists, then ONE pre-pass launch with range + gather blocks.  In the fused
    // step the continuum plane rides with the classification launch — an HBM stream that leaves the vector units idle —
    // instead of the pre-pass launch, which a shard's line blocks already fill (S-c3 / 8: 540 line + gather blocks and 590
    // continuum blocks on 512 block slots)
    bool continuum_done = false;
    if (cull) {
        w.sel = sel;
        // classification blocks: contiguous runs of lines, a few blocks per CU (256 threads: 4 waves x 4 lines x 3 arrays in flight)
        static const int cls_blocks_env = knob("SDX_CLS_BLOCKS") ? std::atoi(knob("SDX_CLS_BLOCKS")) : 0;  // experiment knob
        const int64_t cls_begin = ph && ph->phase == 1 ? ph->begin : 0, cls_end = ph && ph->phase == 1 ? ph->begin + ph->count : n_lines;
        // (a SHARE of the list — two-collective mode — is a few microseconds of streaming: sixteen lines per block, one trip of each
        // wave's loop, so that the launch is as long as one round trip and not as a chain of nine; the whole list is a stream that
        // wants its bytes in flight, not short chains)
        const bool share = ph && ph->phase == 1;
        const int64_t cls_per_block = share ? 16 : 64;
        const unsigned n_cls = cls_blocks_env > 0 ? (unsigned)cls_blocks_env
                                                  : (unsigned)std::max<int64_t>(1, std::min<int64_t>((cls_end - cls_begin + cls_per_block - 1) / cls_per_block, (int64_t)8 * ctx->n_cu));
        // the per-line maxima: doubles behind the integer lists of cnt_ws (8-byte aligned), or the caller's array (two-collective mode)
        double* const m_max = ph ? ph->m_max : (double*)(((uintptr_t)(w.hcount + 16 + 3 * ((size_t)n_lines / 1024 + 8)) + 7) & ~(uintptr_t)7);
        double* const dnu_ws = (double*)ctx->small_ws;
        ContPlan cp;
        static const bool no_ride = knob("SDX_NO_CONT_RIDE") != nullptr;  // A/B knob
        if (job && !no_ride) {
            if ((rc = plan_continuum(kBlock, &cp))) return rc;
            continuum_done = cp.tiled;
        }
        if (ph && ph->phase == 2) {
            // the classification launch ran in phase 1 (sdx_synthesize_classify_dev) and left the grid spacing, `sel` and the
            // continuum plane in this context's scratch — for THIS problem, and no buffer has moved since
            const auto& c = ctx->classified;
            REQUIRE(c.valid && c.n_depth == n_depth && c.n_nu == n_nu && c.nu_begin == nu_begin && c.nu_count == nu_count && c.n_lines == n_lines,
                    "synthesize: options->line_m_max needs sdx_synthesize_classify_dev on this context first, for the same grid, shard and line list");
            REQUIRE(c.generation == ctx->ws_generation, "synthesize: the context's scratch was reallocated between sdx_synthesize_classify_dev and the synthesis");
            REQUIRE(!job || continuum_done, "synthesize: two-collective mode needs the tiled continuum");
            // consumed: a second phase 2 needs a new phase 1 (the resident inputs may have been updated in place since; a recorded
            // graph replays both phases without this host check)
            ctx->classified.valid = false;
        } else {
            REQUIRE(!ph || !job || continuum_done, "synthesize: two-collective mode needs the tiled continuum (at most 4096 bound-free levels)");
            LaunchScope ls(ctx, "k_classify");
            if (continuum_done) {
                // half the classification blocks (16 of 32 wave slots per CU: the stream keeps its bytes in flight), the
                // continuum blocks take the other slots
                const unsigned n_cls_half = share ? n_cls : std::max(1u, n_cls / 2);
                hipLaunchKernelGGL(k_classify_continuum, dim3((unsigned)n_partial + n_cls_half + (unsigned)cp.cont_tiles * cp.cont_rows), dim3(kBlock), cp.shmem, ctx->stream,
                                   n_partial, (int)n_cls_half, n_depth, n_nu, n_lines, dnu_ws, doppler, gammas, gamma_cols, alphas, m_max, nus,
                                   cp.cont_tiles, job->nu_begin, job->nu_count, cp.ca, job->plane, job->nu_count, cp.stage_table, line_nus, nu_begin, nu_count,
                                   sel, cls_begin, cls_end, ctx->far_req);
            } else {
                hipLaunchKernelGGL(k_classify, dim3((unsigned)n_partial + n_cls), dim3(kBlock), 0, ctx->stream, n_partial, n_depth, n_nu, n_lines, dnu_ws,
                                   doppler, gammas, gamma_cols, alphas, m_max, nus, line_nus, nu_begin, nu_count, sel, cls_begin, cls_end, ctx->far_req);
            }
            if (ctx->far_req.count && n_partial > 0) ctx->far_req_done = true;
        }
        if (ph && ph->phase == 1) {
            auto& c = ctx->classified;
            c.valid = true, c.n_depth = n_depth, c.n_nu = n_nu, c.nu_begin = nu_begin, c.nu_count = nu_count, c.n_lines = n_lines;
            c.far_ranges = ctx->far_req_done;
            c.far = far_field_on(ctx, n_nu) && c.far_ranges;
            c.generation = ctx->ws_generation;
            return check_launch("k_classify");
        }
        static const bool no_ticket = knob("SDX_NO_PREPASS_TICKET") != nullptr;  // A/B knob: one block per candidate instead
        // the pre-pass launch that follows draws its items from a counter (k_line_prepass_ticket); the list launch zeroes it
        // (worth it where the candidates are many: at 1e6 lines 57 000 of 62 000 candidate blocks are empty and the launch takes 245
        // instead of 292 us on an eighth of the grid; at 1.5e5 lines 9 500 candidates cost less than the counter's round trips)
        const bool ticket = !no_ticket && !(job && !continuum_done) && !gen && !lo_ref && n_line_blocks >= ctx->prepass_ticket_min_blocks;
        if (ticket) w.ticket = (int*)((char*)ctx->small_ws + 2064);
        {
            LaunchScope ls(ctx, "k_hlist");
            ClassSource cs{};
            cs.m_max = m_max, cs.dnu_partial = dnu_ws, cs.n_partial = n_partial, cs.n_nu = n_nu;
            launch_line_lists(ctx, n_lines, w, pre_lines, &cs);
        }
        w.gather = n_line_blocks;  // worst case: every line listed; blocks beyond the lists' end return at once
        static const bool no_front = knob("SDX_NO_PREPASS_FRONT") != nullptr;  // A/B knob
        w.front = no_front ? 0 : 1;
    }
    const dim3 grid((unsigned)(n_line_blocks + n_pixel_blocks + w.gather), (unsigned)((n_depth + kPreDepths - 1) / kPreDepths));
    if (job && !continuum_done) {
        ContPlan cp;
        if ((rc = plan_continuum(kPreBlock, &cp))) return rc;
        const ContinuumArgs& ca = cp.ca;
        const int cont_tiles = cp.cont_tiles, stage_table = cp.stage_table;
        const size_t shmem = cp.shmem;
        const unsigned total_blocks = grid.x * grid.y + (unsigned)cont_tiles * cp.cont_rows;
        {
        LaunchScope ls(ctx, "k_prepass_continuum");
#define SDX_PRE_ARGS (int)grid.x, (int)grid.y, cont_tiles, n_depth, n_nu, nus, scan_in_block ? (const double*)nullptr : (const double*)ctx->small_ws, \
                     n_partial, n_lines, line_nus, doppler, gammas, gamma_cols, alphas, w, n_line_blocks, job->nu_begin, job->nu_count, ca,          \
                     job->plane, job->nu_count, lp, stage_table
        if (gen && pre_lines == 16) hipLaunchKernelGGL((k_prepass_continuum<true, 16>), dim3(total_blocks), dim3(kPreBlock), shmem, ctx->stream, SDX_PRE_ARGS);
        else if (gen) hipLaunchKernelGGL((k_prepass_continuum<true, 32>), dim3(total_blocks), dim3(kPreBlock), shmem, ctx->stream, SDX_PRE_ARGS);
        else if (pre_lines == 16) hipLaunchKernelGGL((k_prepass_continuum<false, 16>), dim3(total_blocks), dim3(kPreBlock), shmem, ctx->stream, SDX_PRE_ARGS);
        else if (pre_lines == 48) hipLaunchKernelGGL((k_prepass_continuum<false, 48>), dim3(total_blocks), dim3(kPreBlock), shmem, ctx->stream, SDX_PRE_ARGS);
        else if (pre_lines == 54) hipLaunchKernelGGL((k_prepass_continuum<false, 54>), dim3(total_blocks), dim3(kPreBlock), shmem, ctx->stream, SDX_PRE_ARGS);
        else hipLaunchKernelGGL((k_prepass_continuum<false, 32>), dim3(total_blocks), dim3(kPreBlock), shmem, ctx->stream, SDX_PRE_ARGS);
#undef SDX_PRE_ARGS
        }
    } else {
        LaunchScope ls(ctx, job ? "k_prepass_continuum" : "k_line_prepass");
#define SDX_PRE_ARGS n_depth, n_nu, nus, scan_in_block ? (const double*)nullptr : (const double*)ctx->small_ws, n_partial, n_lines, line_nus, doppler, \
                     gammas, gamma_cols, alphas, w, (int*)lo_ref, (int*)hi_ref, n_line_blocks, lp
        if (w.ticket) {
            const dim3 pgrid(std::min<unsigned>(grid.x, 2u * (unsigned)ctx->n_cu), grid.y);
            const double* dnu_arg = scan_in_block ? (const double*)nullptr : (const double*)ctx->small_ws;
#define SDX_TICKET_ARGS n_depth, n_nu, nus, dnu_arg, n_partial, n_lines, line_nus, doppler, gammas, gamma_cols, alphas, w, n_line_blocks, lp
            if (pre_lines == 16) hipLaunchKernelGGL((k_line_prepass_ticket<16>), pgrid, dim3(kPreBlock), 0, ctx->stream, SDX_TICKET_ARGS);
            else if (pre_lines == 48) hipLaunchKernelGGL((k_line_prepass_ticket<48>), pgrid, dim3(kPreBlock), 0, ctx->stream, SDX_TICKET_ARGS);
            else if (pre_lines == 54) hipLaunchKernelGGL((k_line_prepass_ticket<54>), pgrid, dim3(kPreBlock), 0, ctx->stream, SDX_TICKET_ARGS);
            else hipLaunchKernelGGL((k_line_prepass_ticket<32>), pgrid, dim3(kPreBlock), 0, ctx->stream, SDX_TICKET_ARGS);
#undef SDX_TICKET_ARGS
        } else if (pre_lines == 48) hipLaunchKernelGGL((k_line_prepass<false, 48>), grid, dim3(kPreBlock), 0, ctx->stream, SDX_PRE_ARGS);
        else if (pre_lines == 54) hipLaunchKernelGGL((k_line_prepass<false, 54>), grid, dim3(kPreBlock), 0, ctx->stream, SDX_PRE_ARGS);
        else if (gen && pre_lines == 16) hipLaunchKernelGGL((k_line_prepass<true, 16>), grid, dim3(kPreBlock), 0, ctx->stream, SDX_PRE_ARGS);
        else if (gen) hipLaunchKernelGGL((k_line_prepass<true, 32>), grid, dim3(kPreBlock), 0, ctx->stream, SDX_PRE_ARGS);
        else if (pre_lines == 16) hipLaunchKernelGGL((k_line_prepass<false, 16>), grid, dim3(kPreBlock), 0, ctx->stream, SDX_PRE_ARGS);
        else hipLaunchKernelGGL((k_line_prepass<false, 32>), grid, dim3(kPreBlock), 0, ctx->stream, SDX_PRE_ARGS);
#undef SDX_PRE_ARGS
    }
    if (w_out) *w_out = w;
    return check_launch("k_line_prepass");
}

static int check_line_args(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t n_lines,
                           const double* line_nus, const double* doppler, const double* gammas, int gamma_cols,
                           const double* alphas)
{
    REQUIRE(ctx, "null context");
    REQUIRE(n_depth > 0 && n_nu >= 0 && n_lines >= 0, "line opacity: negative or zero sizes");
    REQUIRE(n_nu < (int64_t)2147483647, "line opacity: n_nu must fit int32");
    REQUIRE(n_lines < (int64_t)2147483647, "line opacity: n_lines must fit int32");
    REQUIRE(gamma_cols == n_depth || gamma_cols == 1, "line opacity: gammas must have n_depth or 1 columns");
    REQUIRE(n_nu == 0 || nus, "line opacity: null frequency grid");
    REQUIRE(n_lines == 0 || (line_nus && doppler && gammas && alphas), "line opacity: null line arrays");
    return SDX_OK;
}

// number of line subsets: enough single-wave blocks to fill the chip when the (tile, depth) grid alone is small
// (decided from the GLOBAL grid size, not the shard's: the partition fixes the summation order of every grid point,
// which must not depend on how the grid is sharded)
static int choose_splits(int n_depth, int64_t n_nu_global, int64_t n_lines, int R, int min_splits)
{
    const int64_t tiles = (n_nu_global + 64 * R - 1) / (64 * R);
    const int64_t chunks = (n_lines + 63) / 64;
    int64_t target = 2560;  // S-c2: 2 subsets (4 measured the same, 8 slower: 40.4 / 40.7 / 50.1 us — the narrow role shares the workgroup size)
    if (const char* e = knob("SDX_WIDE_BLOCKS")) target = std::max(1, std::atoi(e));  // tuning knob
    // at least two subsets (four for long lists): the choice must not depend on the shard (it fixes the summation order), and
    // a rank that owns 1/8 of a large grid still needs enough, small enough waves — a (tile, depth) of S-c3 is ~2e4
    // instructions per subset at S = 2, and a shard's last generation of such waves is most of its run time
    const int64_t want = std::max<int64_t>(min_splits, (target + tiles * n_depth - 1) / (tiles * n_depth));
    return (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(want, chunks), 8));  // the subsets are the waves of one workgroup
}

// FAR FIELD (k_line_far): a third plane.  Like every choice that moves a rounding it is made from the GLOBAL grid — grids of at
// least kFarMinPoints frequencies (below that a launch of the line kernel is bound by latency, not by its evaluations) — or set
// explicitly (context option "far_field").
constexpr int64_t kFarMinPoints = 32768;
static_assert(kFarMinPoints == 32768, "sdx_far_field_rule reports this constant");
static bool far_field_on(const sdx_ctx* ctx, int64_t n_nu_global)
{
    static const int far_env = knob("SDX_FAR") ? std::atoi(knob("SDX_FAR")) : -1;  // A/B knob: 0 never, 1 whenever possible
    static const int r_mixed_env = knob("SDX_R_MIXED") ? std::atoi(knob("SDX_R_MIXED")) : 4;
    const int far_mode = ctx->far_field >= 0 ? (int)ctx->far_field : far_env;
    if (ctx->mixed_precision && r_mixed_env == 8) return false;  // (experiment knob: 512-point tiles)
    return far_mode != 0 && (far_mode == 1 || n_nu_global >= kFarMinPoints);
}
// The (ihi, ilo) pair of every GLOBAL tile that holds columns of the launch: asked of the step's grid-spacing launch (launch_dnu, the
// classification launch of a culled shard), which computes them on the side.
static int request_far_ranges(sdx_ctx* ctx, int64_t n_nu, int64_t nu_begin, int64_t nu_count)
{
    int rc = ensure(ctx, &ctx->far_ws, &ctx->far_ws_bytes, (size_t)((n_nu + kFarTile - 1) / kFarTile) * 2 * sizeof(int));
    if (rc) return rc;
    const int64_t t_first = nu_begin / kFarTile, t_last = (nu_begin + nu_count - 1) / kFarTile;
    ctx->far_req = FarReq{(int*)ctx->far_ws, t_first, t_last - t_first + 1};
    ctx->far_req_done = false;
    return SDX_OK;
}

// pre-pass + the two gather kernels; leaves *n_planes_out partial planes in *partial_out ([planes][n_depth][*pld_out]):
// the wide subsets in subset order, then the narrow-window plane.  With `job` the continuum plane of the fused step is
// computed by the same launch as the pre-pass.
static int line_partials(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t nu_begin, int64_t nu_count,
                         int64_t n_lines, const double* line_nus, const double* doppler, const double* gammas, int gamma_cols,
                         const double* alphas, const double** partial_out, int64_t* pld_out, int* n_planes_out, LineWork* w_out,
                         bool count_evals, const ContinuumJob* job = nullptr, const LineParams* gen = nullptr, const ClassifyPhase* ph = nullptr)
{
    constexpr int R = 4;       // grid points per lane of a wide-role tile (tile = 64 R points)
    constexpr int R_MIXED = 4;  // fp32 far wings (8 — twice the points per fetched record — measured slower: fewer tiles qualify as far wing)
    static const int r_mixed_env = knob("SDX_R_MIXED") ? std::atoi(knob("SDX_R_MIXED")) : R_MIXED;  // experiment knob: 4 or 8
    const int Rm = ctx->mixed_precision ? (r_mixed_env == 8 ? 8 : R_MIXED) : R;
    const bool far = far_field_on(ctx, n_nu) && nu_count > 0;
    int rc;
    // (phase 1 computed the ranges — with the far field on then; if the option was switched on in between, k_far_ranges below does it)
    const bool classified_far = ph && ph->phase == 2 && ctx->classified.valid && ctx->classified.far && ctx->classified.far_ranges;
    if (far && !classified_far && (rc = request_far_ranges(ctx, n_nu, nu_begin, nu_count))) return rc;
    LineWork w;
    rc = line_prepass(ctx, n_depth, n_nu, nus, n_lines, line_nus, doppler, gammas, gamma_cols, alphas, true, nullptr, nullptr, &w,
                      count_evals, job, gen, nu_begin, nu_count, ph);
    const FarReq far_req = ctx->far_req;
    const bool far_req_done = ctx->far_req_done || classified_far;
    ctx->far_req = FarReq{nullptr, 0, 0};
    if (rc) return rc;
    const int n_split = choose_splits(n_depth, n_nu, n_lines, Rm, n_lines >= ctx->indexed_min_lines ? 4 : 2);
    // long line lists: the lines with a window wider than kMediumHalfWidth are listed once (they are scanned by every tile);
    // all others are found by centre range.  Short lists are scanned completely.
    const int indexed = n_lines >= ctx->indexed_min_lines ? 1 : 0;
    if (indexed && !w.hlist) {  // (a culled pre-pass has built the lists already)
        LaunchScope ls(ctx, "k_hlist");
        launch_line_lists(ctx, n_lines, w, 32);
    }
    static const bool no_hscan = knob("SDX_NO_HSCAN") != nullptr;  // A/B knob
    if (no_hscan) w.hscan = nullptr;
    if (indexed && !no_hscan && !w.sel) {  // the huge lines' scan words in list order (a culled pre-pass has written them itself)
        LaunchScope ls(ctx, "k_hlist");
        hipLaunchKernelGGL(k_hscan, dim3(64, (unsigned)n_depth), dim3(kBlock), 0, ctx->stream, n_depth, n_lines, (const int*)w.hlist, (const int*)w.hcount,
                           (const WideScan*)w.wscan, w.hscan);
    }
    // two planes: [0] wide windows (the S subsets are summed inside their workgroup), [1] narrow windows.  (Twice the line subsets
    // for the deepest, hottest layers — two workgroups per (depth, tile), a third plane for the second halves — was measured in
    // round 4 at S-c3 and on its eight shards with 0 / 14 / 28 such layers: line kernel 2026 / 2048 / 2092 us unsharded, 332 / 344 /
    // 346 us on the slowest shard, no change at S-c2 either: the hit lists fall off by only a factor 4 from the deepest to the
    // shallowest layer, and a launch is bound by its total work, not by its heaviest wave.  Removed.)
    w.far_range = nullptr;
    if (far) {
        if (!far_req_done) {  // (no grid-spacing launch in this step: small grids scan the spacing inside the pre-pass blocks)
            LaunchScope ls(ctx, "k_far_ranges");
            hipLaunchKernelGGL(k_far_ranges, dim3((unsigned)((2 * far_req.count + kBlock / 16 - 1) / (kBlock / 16))), dim3(kBlock), 0, ctx->stream, n_nu, nus, far_req);
        }
        w.far_range = (const int*)ctx->far_ws;
    }
    rc = ensure(ctx, &ctx->part_ws, &ctx->part_ws_bytes, (size_t)(far ? 3 : 2) * n_depth * nu_count * sizeof(double));
    if (rc) return rc;
    double* part = (double*)ctx->part_ws;
    const int64_t pld = nu_count;
    // tiles are aligned to the GLOBAL grid (multiples of 64 Rm points from index 0), whatever the shard: which points share a
    // tile — and with it how a (line, depth, tile) is classified and which points share a reciprocal — is a property of the grid
    const int tiles = (int)((nu_begin + nu_count + 64 * Rm - 1) / (64 * Rm) - nu_begin / (64 * Rm));
    // order of the wide role's tiles over the XCDs: one contiguous eighth each (0), or groups of g tiles going round them —
    // better balance where an eighth is only a few tiles (a shard's), at the price of fewer neighbouring tiles per L2
    static const int wide_group_env = knob("SDX_WIDE_GROUP") ? std::atoi(knob("SDX_WIDE_GROUP")) & 15 : -1;
    const int wide_group = wide_group_env >= 0 ? wide_group_env : 0;
    const int64_t tiles_pad = wide_group ? ((int64_t)tiles + 8 * wide_group - 1) / (8 * wide_group) * (8 * wide_group) : tiles;
    const int64_t n_wide = tiles_pad * n_depth;
    // workgroups of n_split waves, rounded up to whole rounds of the XCD-aware order (surplus workgroups return at once)
    // narrow role: F consecutive frequencies per wave (a line's records, loaded once, serve F evaluations) for DENSE lists —
    // at least one line per two grid points, where a frequency visits many lines and the walk is bound by its loads and
    // instructions — as long as >= 8192 such waves remain: an eighth of S-c3 (11 000 - 21 000 columns) with F = 4 ran its
    // line kernel 20 % SLOWER (414 against 345 us: a few thousand four-times-longer waves are the launch's tail); sparse lists
    // on small grids (S-c2: 2000 lines on 7634 points, a latency-bound launch) keep one frequency per wave.  Pure scheduling:
    // every frequency adds its lines in the same order whatever F.  (F = 8 was measured slower than 4 at every size.)
    static const int narrow_f_env = knob("SDX_NARROW_F") ? std::atoi(knob("SDX_NARROW_F")) : 0;  // A/B knob: 1, 2, 4
    int narrow_f = 1;
    if (2 * n_lines >= n_nu) narrow_f = nu_count >= 32768 ? 4 : (nu_count >= 16384 ? 2 : 1);
    if (narrow_f_env == 1 || narrow_f_env == 2 || narrow_f_env == 4) narrow_f = narrow_f_env;
    // VERY dense long lists (>= 4 lines per grid point, four line subsets): the four waves of a narrow-role workgroup share one group
    // of four frequencies and split its candidate lines (line_narrow_subsets) — the record sharing of F = 4 with the wave count of
    // F = 1, whatever the width of the launch.  Decided from the global list and grid: it changes the order of a sum.  Measured in
    // round 5: 1e6 lines on 120 398 points (8.3 per point) 10.2 -> 9.9 ms for the whole grid and 1.52 -> 1.36 ms on an eighth; 1.5e5
    // lines (1.25 per point: ~17 lines per wave, less than the wave's start-up and the workgroup's reduction) 2.03 -> 2.20 ms and
    // 311 -> 320 us — hence the density in the rule.
    static const bool no_narrow_subsets = knob("SDX_NO_NARROW_SUBSETS") != nullptr;  // A/B knob
    static const int sub_density_env = knob("SDX_NARROW_SUBSETS_DENSITY") ? std::atoi(knob("SDX_NARROW_SUBSETS_DENSITY")) : -1;  // experiment knob: halves of a line per grid point
    const int64_t sub_density = sub_density_env >= 0 ? sub_density_env : 8;
    const int narrow_sub = (!no_narrow_subsets && 2 * n_lines >= sub_density * n_nu && n_split == 4 && Rm != 8) ? 4 : 0;
    if (narrow_sub) narrow_f = 4;
    const int64_t n_grp = (nu_begin + nu_count + narrow_f - 1) / narrow_f - nu_begin / narrow_f;
    const int64_t n_narrow_units = n_grp * ((n_depth + 63) / 64);
    // (whole rounds of the XCD-aware order: 8 XCDs x groups of 4 workgroups, 16 in the subsets kernel)
    const int64_t narrow_round = narrow_sub ? 128 : 32;
    const int64_t n_narrow = (((narrow_sub ? n_narrow_units : (n_narrow_units + n_split - 1) / n_split) + narrow_round - 1) / narrow_round) * narrow_round;
    static const int narrow_order = knob("SDX_NARROW_ORDER") ? atoi(knob("SDX_NARROW_ORDER")) & 3 : 0;
    REQUIRE(n_wide + n_narrow < ((int64_t)1 << 31), "line opacity: grid too large for one launch");
    static const bool split_launches = knob("SDX_SPLIT_LAUNCHES") != nullptr;  // analysis knob: time the two roles apart
    const bool split_launches_early = split_launches;
    size_t shmem = (size_t)n_split * (far && SDX_WIDE_QUEUED ? kWideFarLdsDoubles : kWideLdsDoubles) * sizeof(double);
    // FAR FIELD: units of 4 RF global tiles; RF (1 or 2 node groups per lane) is scheduling only (4 was measured slower than 2 at every size).  Its workgroups are the FIRST of the line kernel's grid (one launch, and a
    // shard's far waves — the launch's longest chains — run beside the other roles instead of alone on the chip); experiment knob
    // SDX_FAR_LAUNCH gives them a launch of their own, with kFarSplit waves per workgroup.  Either way the number of line subsets is
    // a constant of the mode (it fixes the order of a node's sum): n_split merged, kFarSplit alone.
    static const bool far_own_launch = knob("SDX_FAR_LAUNCH") != nullptr;
    static const int far_rf_env = knob("SDX_FAR_RF") ? std::atoi(knob("SDX_FAR_RF")) : 0;  // experiment knob: 1, 2
    const int64_t far_t_first = nu_begin / (64 * R), far_t_last = (nu_begin + nu_count - 1) / (64 * R);
    auto far_units_of = [&](int f) { return far_t_last / (4 * f) - far_t_first / (4 * f) + 1; };
    // (round 6: units of 8 tiles at every size — on the eighths of S-c3 / S-c4m, where the rule above chose 4, the line launch ran 2 - 5 %
    // faster with 8: half as many waves walk the huge lines' scan words)
    int far_rf = 2;
    if (far_rf_env == 1 || far_rf_env == 2) far_rf = far_rf_env;
    const int64_t far_units = far ? far_units_of(far_rf) : 0;
    const bool far_merged = far && !far_own_launch && !split_launches_early;
    const int64_t n_far = far_merged ? far_units * n_depth : 0;
    if (far_merged) shmem = std::max(shmem, (((size_t)n_split + 2) * far_rf * 64 + (size_t)n_split * kFarWaveLdsDoubles) * sizeof(double));
    REQUIRE(n_far + n_wide + n_narrow < ((int64_t)1 << 31), "line opacity: grid too large for one launch");
    const dim3 g((unsigned)(n_far + n_wide + n_narrow)), blk((unsigned)(64 * n_split));
    // (k_line_far on a second stream beside k_line_all — a fork and a join per step — was measured in round 5: S-c3 1.857 -> 1.840 ms, its
    // eighth 0.419 -> 0.420, S-c4m 7.49 -> 7.48: both kernels are bound by their instructions, neither leaves the other idle slots)
    auto launch_far = [&]() -> int {
        const int rf = far_rf;
        const int64_t units = far_units;
        REQUIRE(units * n_depth < ((int64_t)1 << 31), "line opacity: grid too large for one launch");
        static const int far_split_env = knob("SDX_FAR_SPLIT") ? std::atoi(knob("SDX_FAR_SPLIT")) : 0;  // experiment knob: 1 .. 8
        const int far_split = far_split_env >= 1 && far_split_env <= 8 ? far_split_env : kFarSplit;
        const size_t far_shmem = (((size_t)far_split + 2) * rf * 64 + (size_t)far_split * kFarWaveLdsDoubles) * sizeof(double);
        const dim3 fg((unsigned)(units * n_depth)), fblk((unsigned)(64 * far_split));
        double* far_plane = part + (size_t)2 * n_depth * pld;
        LaunchScope ls(ctx, "k_line_far");
#define SDX_FAR_ARGS (int)units, far_split, n_depth, n_nu, nus, nu_begin, nu_count, n_lines, w, far_plane, pld
        if (rf == 2) hipLaunchKernelGGL((k_line_far<R, 2>), fg, fblk, far_shmem, ctx->stream, SDX_FAR_ARGS);
        else hipLaunchKernelGGL((k_line_far<R, 1>), fg, fblk, far_shmem, ctx->stream, SDX_FAR_ARGS);
#undef SDX_FAR_ARGS
        return SDX_OK;
    };
    for (int pass = 0; pass < (split_launches ? 2 : 1); ++pass) {
        // (raising the priority of the hot layers' waves with s_setprio was measured in round 4: the instruction has side effects as
        // far as the compiler is concerned, the record fetches of the walk stopped being scalar loads and the kernel ran 37 % slower)
        const int roles = (split_launches ? (1 << pass) : 3) | (narrow_order << 2) | (wide_group << 4) | (narrow_f << 8) | (narrow_sub << 12) | ((far_merged ? far_rf : 0) << 16);
        LaunchScope ls(ctx, split_launches ? (pass ? "k_line_narrow" : "k_line_wide") : "k_line_all", far_merged ? "k_line_all + far role" : (far ? "k_line_all (far field in k_line_far)" : nullptr));
#define SDX_LINE_ARGS (int)n_wide, tiles, n_split, n_depth, n_nu, nus, nu_begin, nu_count, n_lines, line_nus, w, part, pld, roles, (int)far_units
        if (ctx->mixed_precision && Rm == 8) hipLaunchKernelGGL((k_line_all_mixed<8>), g, blk, shmem, ctx->stream, SDX_LINE_ARGS);
        else if (ctx->mixed_precision && narrow_sub && far) hipLaunchKernelGGL((k_line_all_mixed<R_MIXED, true, true>), g, blk, shmem, ctx->stream, SDX_LINE_ARGS);
        else if (ctx->mixed_precision && narrow_sub) hipLaunchKernelGGL((k_line_all_mixed<R_MIXED, true>), g, blk, shmem, ctx->stream, SDX_LINE_ARGS);
        else if (ctx->mixed_precision && far) hipLaunchKernelGGL((k_line_all_mixed<R_MIXED, false, true>), g, blk, shmem, ctx->stream, SDX_LINE_ARGS);
        else if (ctx->mixed_precision) hipLaunchKernelGGL((k_line_all_mixed<R_MIXED>), g, blk, shmem, ctx->stream, SDX_LINE_ARGS);
        else if (narrow_sub && far) hipLaunchKernelGGL((k_line_all<R, true, true>), g, blk, shmem, ctx->stream, SDX_LINE_ARGS);
        else if (narrow_sub) hipLaunchKernelGGL((k_line_all<R, true>), g, blk, shmem, ctx->stream, SDX_LINE_ARGS);
        else if (far) hipLaunchKernelGGL((k_line_all<R, false, true>), g, blk, shmem, ctx->stream, SDX_LINE_ARGS);
        else hipLaunchKernelGGL((k_line_all<R>), g, blk, shmem, ctx->stream, SDX_LINE_ARGS);
#undef SDX_LINE_ARGS
    }
    if (far && !far_merged && (rc = launch_far())) return rc;
    *partial_out = part;
    *pld_out = pld;
    *n_planes_out = far ? 3 : 2;
    if (w_out) *w_out = w;
    return check_launch("line kernels");
}

static int line_opacity_impl(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t nu_begin, int64_t nu_count,
                            int64_t n_lines, const double* line_nus, const double* doppler, const double* gammas,
                            int gamma_cols, const double* alphas, double* out, int64_t out_ld, int accumulate,
                            int64_t* n_evaluations_dev, const LineParams* gen)
{
    int rc;
    REQUIRE(nu_begin >= 0 && nu_count >= 0 && nu_begin + nu_count <= n_nu, "line opacity: shard outside the grid");
    REQUIRE(nu_count == 0 || (out && out_ld >= nu_count), "line opacity: bad output buffer");
    if (nu_count == 0) return SDX_OK;
    if (n_lines == 0) {
        if (!accumulate) HIP_TRY(hipMemset2DAsync(out, out_ld * sizeof(double), 0, nu_count * sizeof(double), n_depth, ctx->stream));
        if (n_evaluations_dev) HIP_TRY(hipMemsetAsync(n_evaluations_dev, 0, sizeof(int64_t), ctx->stream));
        return SDX_OK;
    }
    const double* part;
    int64_t pld;
    int n_split;
    LineWork w;
    rc = line_partials(ctx, n_depth, n_nu, nus, nu_begin, nu_count, n_lines, line_nus, doppler, gammas, gamma_cols, alphas, &part, &pld,
                       &n_split, &w, n_evaluations_dev != nullptr, nullptr, gen);
    if (rc) return rc;
    {
        LaunchScope ls(ctx, "k_reduce_partials");
        hipLaunchKernelGGL(k_reduce_partials, grid2(nu_count, n_depth), dim3(kBlock), 0, ctx->stream, n_depth, nu_count, n_split,
                           part, pld, out, out_ld, accumulate);
    }
    rc = check_launch("k_reduce_partials");
    if (rc) return rc;
    if (n_evaluations_dev)
        HIP_TRY(hipMemcpyAsync(n_evaluations_dev, w.evals, sizeof(int64_t), hipMemcpyDeviceToDevice, ctx->stream));
    return SDX_OK;
}

int sdx_line_opacity_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t nu_begin, int64_t nu_count,
                         int64_t n_lines, const double* line_nus, const double* doppler, const double* gammas,
                         int gamma_cols, const double* alphas, double* out, int64_t out_ld, int accumulate,
                         int64_t* n_evaluations_dev)
{
    int rc = check_line_args(ctx, n_depth, n_nu, nus, n_lines, line_nus, doppler, gammas, gamma_cols, alphas);
    if (rc) return rc;
    return line_opacity_impl(ctx, n_depth, n_nu, nus, nu_begin, nu_count, n_lines, line_nus, doppler, gammas, gamma_cols, alphas, out, out_ld,
                             accumulate, n_evaluations_dev, nullptr);
}

// ---- line parameters generated on the device
static int to_line_params(const sdx_linelist* ll, int n_depth, LineParams* lp)
{
    REQUIRE(ll, "line list: null description");
    REQUIRE(ll->n_lines >= 0 && ll->n_lines < (int64_t)2147483647 && n_depth > 0, "line list: bad sizes (n_lines must fit int32)");
    REQUIRE(ll->gamma_mode >= 0 && ll->gamma_mode <= 3, "line list: gamma_mode must be 0..3");
    if (ll->n_lines == 0) return SDX_OK;
    REQUIRE(ll->nu && ll->e_low_ev && ll->strength && ll->pop_row && ll->pop && ll->mass && ll->temperature,
            "line list: null alpha_line / doppler inputs");
    REQUIRE(ll->n_pop_rows > 0, "line list: n_pop_rows must be positive");
    if (ll->gamma_mode <= 1) {
        REQUIRE(ll->atomic_number && ll->ion_number && ll->ionization_energy && ll->upper_energy && ll->lower_energy && ll->A_ul &&
                    ll->electron_density && ll->h_density,
                "line list: null broadening inputs");
        REQUIRE(ll->gamma_mode == 0 || (ll->stark && ll->waals), "line list: VALD broadening needs stark and waals");
    }
    REQUIRE(ll->gamma_mode != 2 || ll->A_ul, "line list: gamma_mode 2 needs A_ul");
    lp->e_low_ev = ll->e_low_ev;
    lp->g_lo = ll->g_lo;
    lp->strength = ll->strength;
    lp->pop_row = (const int*)ll->pop_row;
    lp->pop = ll->pop;
    lp->alpha_coefficient = ll->alpha_coefficient;
    lp->mass = ll->mass;
    lp->xi = ll->microturbulence;
    lp->gamma_mode = ll->gamma_mode;
    lp->flags = ll->broadening_flags;
    lp->z = (const int*)ll->atomic_number;
    lp->ion = (const int*)ll->ion_number;
    lp->e_ion = ll->ionization_energy;
    lp->e_up = ll->upper_energy;
    lp->e_lo = ll->lower_energy;
    lp->a_ul = ll->A_ul;
    lp->stark = ll->stark;
    lp->waals = ll->waals;
    lp->temps = ll->temperature;
    lp->n_e = ll->electron_density;
    lp->n_h = ll->h_density;
    return SDX_OK;
}

int sdx_line_params_dev(sdx_ctx* ctx, int n_depth, const sdx_linelist* ll, double* alphas, double* gammas, double* doppler)
{
    REQUIRE(ctx, "null context");
    LineParams lp{};
    int rc = to_line_params(ll, n_depth, &lp);
    if (rc) return rc;
    if (ll->n_lines == 0) return SDX_OK;
    {
        LaunchScope ls(ctx, "k_line_params");
        hipLaunchKernelGGL(k_line_params, dim3(blocks1(ll->n_lines * n_depth)), dim3(kBlock), 0, ctx->stream, ll->n_lines, n_depth, ll->nu,
                           lp, alphas, gammas, ll->gamma_mode >= 2 ? 1 : n_depth, doppler);
    }
    return check_launch("k_line_params");
}

int sdx_alpha_line_levels_dev(sdx_ctx* ctx, int64_t n_lines, int n_depth, int n_levels, const double* level_density,
                              const int32_t* lower_index, const double* stim, const double* f_lu, double alpha_coefficient,
                              double* alphas)
{
    REQUIRE(ctx && n_lines >= 0 && n_depth > 0 && n_levels >= 0, "alpha_line_levels: bad sizes");
    if (n_lines == 0) return SDX_OK;
    REQUIRE(level_density && lower_index && stim && f_lu && alphas && n_levels > 0, "alpha_line_levels: null pointer");
    {
        LaunchScope ls(ctx, "k_alpha_line_levels");
        hipLaunchKernelGGL(k_alpha_line_levels, dim3(blocks1(n_lines * n_depth)), dim3(kBlock), 0, ctx->stream, n_lines, n_depth, level_density,
                           (const int*)lower_index, stim, f_lu, alpha_coefficient, alphas);
    }
    return check_launch("k_alpha_line_levels");
}

int sdx_line_opacity_linelist_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t nu_begin, int64_t nu_count,
                                  const sdx_linelist* ll, double* out, int64_t out_ld, int accumulate, int64_t* n_evaluations_dev)
{
    REQUIRE(ctx, "null context");
    REQUIRE(n_depth > 0 && n_nu >= 0 && n_nu < (int64_t)2147483647, "line opacity: bad sizes");
    REQUIRE(n_nu == 0 || nus, "line opacity: null frequency grid");
    LineParams lp{};
    int rc = to_line_params(ll, n_depth, &lp);
    if (rc) return rc;
    return line_opacity_impl(ctx, n_depth, n_nu, nus, nu_begin, nu_count, ll->n_lines, ll->nu, nullptr, nullptr,
                             ll->gamma_mode >= 2 ? 1 : n_depth, nullptr, out, out_ld, accumulate, n_evaluations_dev, &lp);
}

int sdx_line_windows_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t n_lines, const double* line_nus,
                         const double* doppler, const double* gammas, int gamma_cols, const double* alphas, int32_t* lower,
                         int32_t* upper)
{
    int rc = check_line_args(ctx, n_depth, n_nu, nus, n_lines, line_nus, doppler, gammas, gamma_cols, alphas);
    if (rc) return rc;
    REQUIRE(n_lines == 0 || (lower && upper), "line windows: null output");
    if (n_lines == 0) return SDX_OK;
    return line_prepass(ctx, n_depth, n_nu, nus, n_lines, line_nus, doppler, gammas, gamma_cols, alphas, false, lower, upper,
                        nullptr);
}

// Host-pointer (*_f64) entry points — what a numpy caller (the reference's calc_alpha_line_at_nu, base.py:432-439) binds.
// Their device buffers are sub-allocated from ONE arena owned by the context and their transfers go through ONE pinned host
// buffer, both grown on demand and kept across calls: a call costs no hipMalloc / hipFree and every copy is a DMA from / to
// page-locked memory.  Inputs: memcpy user -> pinned, async DMA pinned -> device (the DMA of array k overlaps the memcpy of
// array k + 1).  Outputs: async DMA device -> pinned, one synchronize, memcpy pinned -> user.  Transfers beyond
// kPinnedStagingMax go straight from / to the caller's pages (the runtime pins them on the fly).
constexpr size_t kPinnedStagingMax = (size_t)512 << 20;

struct HostIo {
    sdx_ctx* ctx;
    size_t dev_off = 0, pin_off = 0;
    bool pinned = true;
    struct Pending {
        void* dst;
        const void* src_pin;
        size_t bytes;      // bytes per row
        size_t rows = 1;   // rows of `bytes`, packed in the pinned buffer, dst_pitch apart in the destination
        size_t dst_pitch = 0;
    };
    std::vector<Pending> pending;
    bool finished = false;

    explicit HostIo(sdx_ctx* c) : ctx(c) {}
    HostIo(const HostIo&) = delete;
    HostIo& operator=(const HostIo&) = delete;
    // error exits return before finish(): DMAs out of / into the context's pinned buffer may still be in flight, and the next
    // call would memcpy new data over them
    ~HostIo()
    {
        if (!finished) hipStreamSynchronize(ctx->stream);
    }

    static size_t pad(size_t b) { return (b + 255) & ~(size_t)255; }

    // sizes: total device bytes and total staged host bytes of the call (padded per array by the caller through need())
    int begin(size_t dev_need, size_t pin_need)
    {
        static const bool no_pin = knob("SDX_NO_PINNED_STAGING") != nullptr;
        pinned = !no_pin && pin_need <= kPinnedStagingMax;
        if (ctx->io_dev_bytes < dev_need) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (ctx->io_dev) HIP_TRY(hipFree(ctx->io_dev));
            ctx->io_dev = nullptr;
            ctx->io_dev_bytes = 0;
            const size_t want = dev_need + dev_need / 4;  // head-room: repeated calls with slowly growing sizes do not reallocate each time
            HIP_TRY(hipMalloc(&ctx->io_dev, want));
            ctx->io_dev_bytes = want;
        }
        if (pinned && ctx->io_pin_bytes < pin_need) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (ctx->io_pin) HIP_TRY(hipHostFree(ctx->io_pin));
            ctx->io_pin = nullptr;
            ctx->io_pin_bytes = 0;
            const size_t want = pin_need + pin_need / 4;
            HIP_TRY(hipHostMalloc(&ctx->io_pin, want, hipHostMallocDefault));
            ctx->io_pin_bytes = want;
        }
        return SDX_OK;
    }
    void* alloc(size_t bytes)
    {
        void* p = (char*)ctx->io_dev + dev_off;
        dev_off += pad(bytes ? bytes : 8);
        return p;
    }
    int upload(const void* src, size_t bytes, const void** dev_out)
    {
        void* d = alloc(bytes);
        *dev_out = d;
        if (!bytes) return SDX_OK;
        const void* from = src;
        if (pinned) {
            void* h = (char*)ctx->io_pin + pin_off;
            pin_off += pad(bytes);
            std::memcpy(h, src, bytes);
            from = h;
        }
        HIP_TRY(hipMemcpyAsync(d, from, bytes, hipMemcpyHostToDevice, ctx->stream));
        return SDX_OK;
    }
    int download(void* dst, const void* dev_src, size_t bytes)
    {
        if (!bytes) return SDX_OK;
        if (pinned) {
            void* h = (char*)ctx->io_pin + pin_off;
            pin_off += pad(bytes);
            HIP_TRY(hipMemcpyAsync(h, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
            pending.push_back({dst, h, bytes, 1, 0});
        } else {
            HIP_TRY(hipMemcpyAsync(dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        }
        return SDX_OK;
    }
    // rows x row_bytes, contiguous on the device, into a host array whose rows are dst_pitch bytes apart
    int download2d(void* dst, size_t dst_pitch, const void* dev_src, size_t row_bytes, size_t rows)
    {
        if (dst_pitch == row_bytes || rows <= 1) return download(dst, dev_src, row_bytes * rows);
        if (!row_bytes) return SDX_OK;
        if (pinned) {
            void* h = (char*)ctx->io_pin + pin_off;
            pin_off += pad(row_bytes * rows);
            HIP_TRY(hipMemcpyAsync(h, dev_src, row_bytes * rows, hipMemcpyDeviceToHost, ctx->stream));
            pending.push_back({dst, h, row_bytes, rows, dst_pitch});
        } else {
            HIP_TRY(hipMemcpy2DAsync(dst, dst_pitch, dev_src, row_bytes, row_bytes, rows, hipMemcpyDeviceToHost, ctx->stream));
        }
        return SDX_OK;
    }
    int finish()
    {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        finished = true;
        for (auto& p : pending)
            for (size_t r = 0; r < p.rows; ++r) std::memcpy((char*)p.dst + r * p.dst_pitch, (const char*)p.src_pin + r * p.bytes, p.bytes);
        pending.clear();
        return SDX_OK;
    }
};

static int host_grid_check(int64_t n_nu, const double* nus)
{
    for (int64_t i = 0; i + 1 < n_nu; ++i)
        if (!(nus[i + 1] < nus[i])) return fail(SDX_ERR_ARG, "tracing frequencies must be strictly descending (stardis/base.py:34)");
    return SDX_OK;
}

// The kernels take the line list in ascending frequency (calc_alpha_line_at_nu sorts it, base.py:392-397): cnt_ge is a
// binary search over line_nus and the narrow role walks a contiguous index range per frequency.
static int host_lines_check(int64_t n_lines, const double* line_nus)
{
    for (int64_t l = 0; l + 1 < n_lines; ++l)
        if (!(line_nus[l + 1] >= line_nus[l]))
            return fail(SDX_ERR_ARG, "line_nus must be ascending (sort the line list by frequency, opacities_solvers/base.py:392-397)");
    return SDX_OK;
}

int sdx_line_opacity_f64(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t n_lines, const double* line_nus,
                         const double* doppler, const double* gammas, int gamma_cols, const double* alphas, double* out,
                         int64_t* n_evaluations)
{
    int rc = check_line_args(ctx, n_depth, n_nu, nus, n_lines, line_nus, doppler, gammas, gamma_cols, alphas);
    if (rc) return rc;
    REQUIRE(n_nu == 0 || out, "line opacity: null output");
    if ((rc = host_grid_check(n_nu, nus))) return rc;
    if ((rc = host_lines_check(n_lines, line_nus))) return rc;
    for (int64_t k = 0; k < n_lines * n_depth; ++k)
        if (doppler[k] == 0.0) return fail(SDX_ERR_ARG, "doppler_width == 0 (ZeroDivisionError in the reference, voigt.py:148)");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t f8 = sizeof(double), ld = (size_t)n_lines * n_depth * f8, plane = (size_t)n_depth * n_nu * f8;
    const size_t in_bytes[] = {(size_t)n_nu * f8, (size_t)n_lines * f8, ld, (size_t)n_lines * gamma_cols * f8, ld};
    size_t dev_need = HostIo::pad(plane) + 256, pin_need = HostIo::pad(plane) + 256;
    for (size_t b : in_bytes) dev_need += HostIo::pad(b ? b : 8), pin_need += HostIo::pad(b);
    HostIo io{ctx};
    if ((rc = io.begin(dev_need, pin_need))) return rc;
    const double *d_nus, *d_ln, *d_dw, *d_g, *d_a;
    if ((rc = io.upload(nus, in_bytes[0], (const void**)&d_nus)) || (rc = io.upload(line_nus, in_bytes[1], (const void**)&d_ln)) ||
        (rc = io.upload(doppler, in_bytes[2], (const void**)&d_dw)) || (rc = io.upload(gammas, in_bytes[3], (const void**)&d_g)) ||
        (rc = io.upload(alphas, in_bytes[4], (const void**)&d_a)))
        return rc;
    double* d_out = (double*)io.alloc(plane);
    int64_t* d_ev = (int64_t*)io.alloc(sizeof(int64_t));
    rc = sdx_line_opacity_dev(ctx, n_depth, n_nu, d_nus, 0, n_nu, n_lines, d_ln, d_dw, d_g, gamma_cols, d_a, d_out, n_nu, 0, d_ev);
    if (rc) return rc;
    int64_t ev = 0;
    if ((rc = io.download(out, d_out, plane))) return rc;
    HIP_TRY(hipMemcpyAsync(&ev, d_ev, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = io.finish())) return rc;
    if (n_evaluations) *n_evaluations = ev;
    return SDX_OK;
}

int sdx_faddeeva_dev(sdx_ctx* ctx, int64_t n, const double* z, double* w)
{
    REQUIRE(ctx && n >= 0 && (n == 0 || (z && w)), "faddeeva: bad arguments");
    if (n == 0) return SDX_OK;
    {
        LaunchScope ls(ctx, "k_faddeeva");
        hipLaunchKernelGGL(k_faddeeva, dim3(blocks1(n)), dim3(kBlock), 0, ctx->stream, n, z, w);
    }
    return check_launch("k_faddeeva");
}

int sdx_voigt_profile_dev(sdx_ctx* ctx, int64_t n, const double* dnu, const double* dw, const double* gamma, double* phi)
{
    REQUIRE(ctx && n >= 0 && (n == 0 || (dnu && dw && gamma && phi)), "voigt_profile: bad arguments");
    if (n == 0) return SDX_OK;
    {
        LaunchScope ls(ctx, "k_voigt_profile");
        hipLaunchKernelGGL(k_voigt_profile, dim3(blocks1(n)), dim3(kBlock), 0, ctx->stream, n, dnu, dw, gamma, phi);
    }
    return check_launch("k_voigt_profile");
}

int sdx_voigt_term_dev(sdx_ctx* ctx, int64_t n, const double* delta_nu, const double* inv_doppler_width, const double* y,
                       const double* amp, double* out)
{
    REQUIRE(ctx && n >= 0 && (n == 0 || (delta_nu && inv_doppler_width && y && amp && out)), "voigt_term: bad arguments");
    if (n == 0) return SDX_OK;
    {
        LaunchScope ls(ctx, "k_voigt_term");
        hipLaunchKernelGGL(k_voigt_term, dim3(blocks1(n)), dim3(kBlock), 0, ctx->stream, n, delta_nu, inv_doppler_width, y, amp, out);
    }
    return check_launch("k_voigt_term");
}

int sdx_voigt_term_f32_dev(sdx_ctx* ctx, int64_t n, const double* delta_nu, const double* inv_doppler_width, const double* y,
                           const double* amp, double* out)
{
    REQUIRE(ctx && n >= 0 && (n == 0 || (delta_nu && inv_doppler_width && y && amp && out)), "voigt_term_f32: bad arguments");
    if (n == 0) return SDX_OK;
    {
        LaunchScope ls(ctx, "k_voigt_term32");
        hipLaunchKernelGGL(k_voigt_term32, dim3(blocks1(n)), dim3(kBlock), 0, ctx->stream, n, delta_nu, inv_doppler_width, y, amp, out);
    }
    return check_launch("k_voigt_term32");
}

// ================================================================================================ broadening
int sdx_calc_gamma_dev(sdx_ctx* ctx, int64_t n_lines, int n_depth, const int32_t* z, const int32_t* ion, const double* e_ion,
                       const double* e_up, const double* e_lo, const double* a_ul, const double* ne, const double* temps,
                       const double* nh, int flags, double* gammas)
{
    REQUIRE(ctx && n_lines >= 0 && n_depth > 0, "calc_gamma: bad sizes");
    if (n_lines == 0) return SDX_OK;
    REQUIRE(z && ion && e_ion && e_up && e_lo && a_ul && ne && temps && nh && gammas, "calc_gamma: null pointer");
    {
        LaunchScope ls(ctx, "k_calc_gamma");
        hipLaunchKernelGGL(k_calc_gamma, dim3(blocks1(n_lines * n_depth)), dim3(kBlock), 0, ctx->stream, n_lines, n_depth,
                           (const int*)z, (const int*)ion, e_ion, e_up, e_lo, a_ul, ne, temps, nh, flags, gammas);
    }
    return check_launch("k_calc_gamma");
}

int sdx_doppler_widths_dev(sdx_ctx* ctx, int64_t n_lines, int n_depth, const double* line_nus, const double* mass,
                           const double* temps, double microturbulence, double* out)
{
    REQUIRE(ctx && n_lines >= 0 && n_depth > 0, "doppler_widths: bad sizes");
    if (n_lines == 0) return SDX_OK;
    REQUIRE(line_nus && mass && temps && out, "doppler_widths: null pointer");
    {
        LaunchScope ls(ctx, "k_doppler_widths");
        hipLaunchKernelGGL(k_doppler_widths, dim3(blocks1(n_lines * n_depth)), dim3(kBlock), 0, ctx->stream, n_lines, n_depth,
                           line_nus, mass, temps, microturbulence, out);
    }
    return check_launch("k_doppler_widths");
}

int sdx_calc_vald_gamma_dev(sdx_ctx* ctx, int64_t n_lines, int n_depth, const int32_t* z, const int32_t* ion,
                            const double* e_ion, const double* e_up, const double* e_lo, const double* a_ul,
                            const double* stark, const double* waals, const double* mass, const double* ne,
                            const double* temps, const double* nh, int flags, double* gammas)
{
    REQUIRE(ctx && n_lines >= 0 && n_depth > 0, "calc_vald_gamma: bad sizes");
    if (n_lines == 0) return SDX_OK;
    REQUIRE(z && ion && e_ion && e_up && e_lo && a_ul && stark && waals && mass && ne && temps && nh && gammas,
            "calc_vald_gamma: null pointer");
    {
        LaunchScope ls(ctx, "k_calc_vald_gamma");
        hipLaunchKernelGGL(k_calc_vald_gamma, dim3(blocks1(n_lines * n_depth)), dim3(kBlock), 0, ctx->stream, n_lines,
                           n_depth, (const int*)z, (const int*)ion, e_ion, e_up, e_lo, a_ul, stark, waals, mass, ne, temps, nh,
                           flags, gammas);
    }
    return check_launch("k_calc_vald_gamma");
}

int sdx_broadening_scalar_dev(sdx_ctx* ctx, int op, int64_t n, const double* a, const double* b, const double* c,
                              const double* d, const double* e, double* out)
{
    REQUIRE(ctx && op >= 0 && op <= 4 && n >= 0, "broadening_scalar: bad arguments");
    if (n == 0) return SDX_OK;
    REQUIRE(a && b && c && out && (op == kOpNEff || op == kOpLinearStark || d) && (op < kOpQuadraticStark || e),
            "broadening_scalar: null pointer");
    {
        LaunchScope ls(ctx, "k_broadening_scalar");
        hipLaunchKernelGGL(k_broadening_scalar, dim3(blocks1(n)), dim3(kBlock), 0, ctx->stream, op, n, a, b, c, d, e, out);
    }
    return check_launch("k_broadening_scalar");
}

// ================================================================================================ continuum
static int launch_source(sdx_ctx* ctx, const char* name, int src, int n_depth, int64_t n_nu, const double* nus,
                         const ContinuumArgs& a, double* out, int64_t ld)
{
    REQUIRE(n_depth > 0 && n_nu >= 0 && (n_nu == 0 || (out && ld >= n_nu)), "continuum: bad output");
    if (n_nu == 0) return SDX_OK;
    {
        LaunchScope ls(ctx, name);
        hipLaunchKernelGGL(k_continuum_source, grid2(n_nu, n_depth), dim3(kBlock), 0, ctx->stream, src, n_depth, n_nu, nus, a,
                           out, ld);
    }
    return check_launch(name);
}

int sdx_alpha_file_1d_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* lambdas, int n_table, const double* tab_x,
                          const double* tab_y, const double* density, double* out, int64_t ld)
{
    REQUIRE(ctx && lambdas && tab_x && tab_y && density && n_table > 0, "alpha_file_1d: bad arguments");
    ContinuumArgs a{};
    a.lambdas = lambdas;
    a.n_table = n_table;
    a.table_wavelength = tab_x;
    a.table_sigma = tab_y;
    a.table_density = density;
    return launch_source(ctx, "k_alpha_file_1d", kSrcFile1d, n_depth, n_nu, nullptr, a, out, ld);
}

int sdx_alpha_file_2d_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* sigma, int64_t sigma_ld,
                          const double* density, double* out, int64_t ld)
{
    REQUIRE(ctx && n_depth > 0 && n_nu >= 0 && sigma && density && out && sigma_ld >= n_nu && ld >= n_nu,
            "alpha_file_2d: bad arguments");
    if (n_nu == 0) return SDX_OK;
    {
        LaunchScope ls(ctx, "k_scale_rows");
        hipLaunchKernelGGL(k_scale_rows, grid2(n_nu, n_depth), dim3(kBlock), 0, ctx->stream, n_depth, n_nu, sigma, sigma_ld,
                           density, out, ld);
    }
    return check_launch("k_scale_rows");
}

int sdx_sigma_table_2d_dev(sdx_ctx* ctx, int n_x, const double* x_axis, int n_y, const double* y_axis, const int32_t* cell_simplices,
                           const double* transform, const double* simplex_values, int n_depth, int64_t n_nu, const double* lambdas,
                           const double* second, int scale_kind, const double* temperature, double* sigma, int64_t ld,
                           int32_t* zero_rows)
{
    REQUIRE(ctx && n_x >= 2 && n_y >= 2 && n_x <= kTableAxisMax, "sigma_table_2d: table axes must have 2..512 nodes");
    REQUIRE(x_axis && y_axis && cell_simplices && transform && simplex_values, "sigma_table_2d: null table");
    REQUIRE(n_depth > 0 && n_nu >= 0 && scale_kind >= 0 && scale_kind <= 2, "sigma_table_2d: bad sizes");
    REQUIRE(scale_kind != 2 || temperature, "sigma_table_2d: scale_kind 2 needs the temperatures");
    if (n_nu == 0) return SDX_OK;
    REQUIRE(lambdas && second && sigma && ld >= n_nu, "sigma_table_2d: bad query / output");
    if (zero_rows) HIP_TRY(hipMemsetAsync(zero_rows, 0, (size_t)n_depth * sizeof(int32_t), ctx->stream));
    {
        LaunchScope ls(ctx, "k_sigma_table_2d");
        hipLaunchKernelGGL(k_sigma_table_2d, grid2(n_nu, n_depth), dim3(kBlock), 0, ctx->stream, n_x, x_axis, n_y, y_axis,
                           (const int*)cell_simplices, transform, simplex_values, n_depth, n_nu, lambdas, second, scale_kind, temperature,
                           sigma, ld, (int*)zero_rows);
    }
    return check_launch("k_sigma_table_2d");
}

int sdx_alpha_bf_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int n_species, const int32_t* offs,
                     const int32_t* ions, const double* cutoff, const double* level_density, double* out, int64_t ld)
{
    REQUIRE(ctx && nus && n_species >= 0, "alpha_bf: bad arguments");
    ContinuumArgs a{};
    if (n_species > 0) {
        REQUIRE(offs && ions && cutoff && level_density, "alpha_bf: null pointer");
        int32_t n_levels = 0;  // offsets live on the device; the last one is the level count
        HIP_TRY(hipMemcpyAsync(&n_levels, offs + n_species, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        double* coef = nullptr;
        if (n_levels > 0) {
            int rc = launch_bf_coef(ctx, n_depth, n_species, n_levels, offs, ions, cutoff, level_density, &coef);
            if (rc) return rc;
        }
        a.bf_n_species = n_species;
        a.bf_species_offsets = (const int*)offs;
        a.bf_species_ion_number = (const int*)ions;
        a.bf_cutoff = cutoff;
        a.bf_coef = coef;
    }
    return launch_source(ctx, "k_alpha_bf", kSrcBf, n_depth, n_nu, nus, a, out, ld);
}

int sdx_alpha_ff_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, const double* temps, int n_species,
                     const int32_t* ions, const double* number_density, double* out, int64_t ld)
{
    REQUIRE(ctx && nus && temps && n_species >= 0 && (n_species == 0 || (ions && number_density)), "alpha_ff: bad arguments");
    ContinuumArgs a{};
    a.ff_n_species = n_species;
    a.ff_species_ion_number = (const int*)ions;
    a.ff_number_density = number_density;
    a.temperature = temps;
    return launch_source(ctx, "k_alpha_ff", kSrcFf, n_depth, n_nu, nus, a, out, ld);
}

int sdx_alpha_rayleigh_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, double* nus, const double* n_h, const double* n_he,
                           const double* n_h2, double* out, int64_t ld)
{
    REQUIRE(ctx && nus, "alpha_rayleigh: bad arguments");
    if (n_nu == 0) return SDX_OK;
    {
        LaunchScope ls(ctx, "k_rayleigh_clip");
        hipLaunchKernelGGL(k_rayleigh_clip, dim3(blocks1(n_nu)), dim3(kBlock), 0, ctx->stream, n_nu, nus);
    }
    int rc = check_launch("k_rayleigh_clip");
    if (rc) return rc;
    ContinuumArgs a{};
    a.ray_n_h = n_h;
    a.ray_n_he = n_he;
    a.ray_n_h2 = n_h2;
    return launch_source(ctx, "k_alpha_rayleigh", kSrcRayleigh, n_depth, n_nu, nus, a, out, ld);
}

int sdx_alpha_electron_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* ne, double* out, int64_t ld)
{
    REQUIRE(ctx && ne, "alpha_electron: bad arguments");
    ContinuumArgs a{};
    a.electron_density = ne;
    return launch_source(ctx, "k_alpha_electron", kSrcElectron, n_depth, n_nu, nullptr, a, out, ld);
}

int sdx_accumulate_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, double* total, int64_t tld, const double* src, int64_t sld)
{
    REQUIRE(ctx && n_depth > 0 && n_nu >= 0 && (n_nu == 0 || (total && src && tld >= n_nu && sld >= n_nu)),
            "accumulate: bad arguments");
    if (n_nu == 0) return SDX_OK;
    {
        LaunchScope ls(ctx, "k_accumulate");
        hipLaunchKernelGGL(k_accumulate, grid2(n_nu, n_depth), dim3(kBlock), 0, ctx->stream, n_depth, n_nu, total, tld, src, sld);
    }
    return check_launch("k_accumulate");
}

static int launch_total(sdx_ctx* ctx, int n_depth, const double* nus, int64_t nu_begin, int64_t nu_count,
                        const sdx_continuum* cont, const double* line, int64_t line_ld, int n_split, double* line_out,
                        int64_t line_out_ld, double* total, int64_t total_ld, hipStream_t stream = nullptr)
{
    if (!stream) stream = ctx->stream;
    ContinuumArgs a = to_args(cont, nullptr);
    a.bf_level_density = cont->bf_level_density;
    size_t shmem = 8;
    if (a.bf_n_species > 0) {
        REQUIRE(cont->bf_n_levels > 0 && cont->bf_n_levels <= 4096, "total_alphas: bf_n_levels must be set (1..4096)");
        shmem = (size_t)cont->bf_n_levels * sizeof(double);
    }
    {
        LaunchScope ls(ctx, "k_total_alphas");
        hipLaunchKernelGGL(k_total_alphas, grid2(nu_count, n_depth), dim3(kBlock), shmem, stream, n_depth, nu_begin, nu_count,
                           nus, a, line, line_ld, n_split, line_out, line_out_ld, total, total_ld);
    }
    return check_launch("k_total_alphas");
}

int sdx_total_alphas_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t nu_begin, int64_t nu_count,
                         const sdx_continuum* cont, const double* alpha_line, int64_t line_ld, double* total, int64_t total_ld)
{
    REQUIRE(ctx && cont && nus && n_depth > 0, "total_alphas: bad arguments");
    REQUIRE(nu_begin >= 0 && nu_count >= 0 && nu_begin + nu_count <= n_nu, "total_alphas: shard outside the grid");
    REQUIRE(nu_count == 0 || (total && total_ld >= nu_count && (!alpha_line || line_ld >= nu_count)), "total_alphas: bad buffers");
    if (int rc = check_file_planes(cont, n_nu)) return rc;
    if (nu_count == 0) return SDX_OK;
    return launch_total(ctx, n_depth, nus, nu_begin, nu_count, cont, alpha_line, line_ld, 1, nullptr, 0, total, total_ld);
}

// ================================================================================================ formal solution
int sdx_blackbody_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, const double* temps, double* out, int64_t ld)
{
    REQUIRE(ctx && n_depth > 0 && n_nu >= 0 && (n_nu == 0 || (nus && temps && out && ld >= n_nu)), "blackbody: bad arguments");
    if (n_nu == 0) return SDX_OK;
    {
        LaunchScope ls(ctx, "k_blackbody");
        hipLaunchKernelGGL(k_blackbody, grid2(n_nu, n_depth), dim3(kBlock), 0, ctx->stream, n_depth, n_nu, nus, temps, out, ld);
    }
    return check_launch("k_blackbody");
}

int sdx_calc_weights_dev(sdx_ctx* ctx, int64_t n, const double* tau, double* w0, double* w1, double* w2)
{
    REQUIRE(ctx && n >= 0 && (n == 0 || (tau && w0 && w1 && w2)), "calc_weights: bad arguments");
    if (n == 0) return SDX_OK;
    {
        LaunchScope ls(ctx, "k_weights");
        hipLaunchKernelGGL(k_weights, dim3(blocks1(n)), dim3(kBlock), 0, ctx->stream, n, tau, w0, w1, w2);
    }
    return check_launch("k_weights");
}

static int raytrace_impl(sdx_ctx* ctx, int n_depth, int64_t n_nu, int n_theta, const double* nus, const double* temps,
                         const double* ray_dist, const double* wts, const double* alphas, int64_t ald, double* F, int64_t fld,
                         double* I_nus, int accumulate, int inward, const FusedTotal* fused = nullptr, int64_t nu_global = -1);

int sdx_raytrace_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, int n_theta, const double* nus, const double* temps,
                     const double* ray_dist, const double* wts, const double* alphas, int64_t ald, double* F, int64_t fld,
                     double* I_nus, int accumulate)
{
    return raytrace_impl(ctx, n_depth, n_nu, n_theta, nus, temps, ray_dist, wts, alphas, ald, F, fld, I_nus, accumulate, 0);
}

int sdx_raytrace_source_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, int n_theta, const double* nus, const double* temps,
                            const double* ray_dist, const double* wts, const double* alphas, int64_t ald, const double* source,
                            int64_t source_ld, double* F, int64_t fld, double* I_nus, int accumulate, int inward,
                            double photospheric_correction)
{
    REQUIRE(ctx && (n_nu == 0 || !source || source_ld >= n_nu), "raytrace: bad source plane");
    FusedTotal ft{};
    ft.source = source;
    ft.sld = source_ld;
    int rc = raytrace_impl(ctx, n_depth, n_nu, n_theta, nus, temps, ray_dist, wts, alphas, ald, F, fld, I_nus, accumulate, inward ? 1 : 0,
                           source ? &ft : nullptr);
    if (rc || !inward || !F || n_nu == 0) return rc;
    {
        LaunchScope ls(ctx, "k_scale");
        hipLaunchKernelGGL(k_scale, grid2(n_nu, n_depth), dim3(kBlock), 0, ctx->stream, n_depth, n_nu, F, fld, photospheric_correction);
    }
    return check_launch("k_scale");
}

int sdx_raytrace_spherical_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, int n_theta, const double* nus, const double* temps,
                               const double* ray_dist, const double* wts, const double* alphas, int64_t ald, double* F,
                               int64_t fld, double* I_nus, int accumulate, double photospheric_correction)
{
    int rc = raytrace_impl(ctx, n_depth, n_nu, n_theta, nus, temps, ray_dist, wts, alphas, ald, F, fld, I_nus, accumulate, 1);
    if (rc || !F || n_nu == 0) return rc;
    {
        LaunchScope ls(ctx, "k_scale");
        hipLaunchKernelGGL(k_scale, grid2(n_nu, n_depth), dim3(kBlock), 0, ctx->stream, n_depth, n_nu, F, fld, photospheric_correction);
    }
    return check_launch("k_scale");
}

static int raytrace_impl(sdx_ctx* ctx, int n_depth, int64_t n_nu, int n_theta, const double* nus, const double* temps,
                         const double* ray_dist, const double* wts, const double* alphas, int64_t ald, double* F, int64_t fld,
                         double* I_nus, int accumulate, int inward, const FusedTotal* fused, int64_t nu_global)
{
    if (nu_global < n_nu) nu_global = n_nu;  // stand-alone calls: the grid handed over is the whole grid
    FusedTotal ft{};
    if (fused) ft = *fused;
    REQUIRE(ctx && n_depth >= 2 && n_nu >= 0 && n_theta > 0, "raytrace: need n_depth >= 2, n_theta > 0");
    if (n_nu == 0) return SDX_OK;
    REQUIRE(nus && temps && ray_dist && wts && ((alphas && ald >= n_nu) || ft.cont), "raytrace: null pointer");
    REQUIRE(!ft.cont || n_theta <= 64, "raytrace: the fused total needs all angles in one launch");
    REQUIRE(!ft.source || n_theta <= 64, "raytrace: a caller-provided source plane needs all angles in one launch");
    REQUIRE((F && fld >= n_nu) || I_nus, "raytrace: no output requested");
    constexpr int kMaxChunk = 64;
    for (int th0 = 0; th0 < n_theta; th0 += kMaxChunk) {
        const int nth = std::min(kMaxChunk, n_theta - th0);
        const double* rd = ray_dist + th0;
        const double* w = wts + th0;
        double* inus = I_nus ? I_nus + th0 : nullptr;
        const int acc = (accumulate || th0 > 0) ? 1 : 0;
        // Angles per lane P and lanes per frequency G = ceil(n_theta / P).  P = 1 (one lane per (frequency, angle)) is the
        // default at every size measured; SDX_RT_P overrides it for experiments.
        int P = 1;  // measured on MI355X at 7.6e3 and 1.2e5 frequencies: one angle per lane wins (more waves in flight)
        if (const char* e = knob("SDX_RT_P")) {  // tuning knob: angles per lane (1, 2 or 4)
            const int v = std::atoi(e);
            if (v == 1 || v == 2 || v == 4) P = v;
        }
        const int G = (nth + P - 1) / P;
        const int kbatch = P == 1 ? 4 : 2;
        auto lds_bytes = [&](int groups) {  // per wave: source and sqrt(alpha) columns, flux terms of a batch of gaps
            return ((size_t)(kRtBlock / 64) * (2 * (size_t)groups * n_depth + (size_t)kbatch * groups * P * G)) * sizeof(double);
        };
        int gpw = 64 / G;  // frequencies per wave; lowered (idle lanes) until the staged columns fit 64 KB of LDS
        while (gpw > 1 && lds_bytes(gpw) > 64 * 1024) --gpw;
        const size_t shmem = lds_bytes(gpw);
        const unsigned blocks = (unsigned)((n_nu + (int64_t)gpw * (kRtBlock / 64) - 1) / ((int64_t)gpw * (kRtBlock / 64)));
        const unsigned blocks_basic = (unsigned)((n_nu + (int64_t)(64 / G) * (kBlock / 64) - 1) / ((int64_t)(64 / G) * (kBlock / 64)));
        // plane-parallel, one angle per lane, nothing to add to, a small grid: the gaps of a ray split over the 8 waves of a
        // workgroup (k_raytrace_seg)
        const int seg_gpw = 64 / nth;
        const size_t seg_doubles = seg_lds_doubles(n_depth, nth);
        if (use_segmented_raytrace(ctx, n_depth, nu_global, n_theta, P == 1 && !inward && !acc)) {
            {
                LaunchScope ls(ctx, "k_raytrace", kSegWaves == 4 ? "k_raytrace_seg<4,14>" : "k_raytrace_seg<8,7>");
                const unsigned seg_blocks = (unsigned)(((n_nu + seg_gpw - 1) / seg_gpw + 7) / 8 * 8);  // whole rounds of the XCD-aware order
#define SDX_SEG_ARGS n_depth, n_nu, nth, n_theta, nus, temps, rd, w, alphas, ald, F, fld, inus, seg_gpw, ft
                if (kSegWaves == 4) hipLaunchKernelGGL((k_raytrace_seg<4, 14>), dim3(seg_blocks), dim3(256), seg_doubles * sizeof(double), ctx->stream, SDX_SEG_ARGS);
                else hipLaunchKernelGGL((k_raytrace_seg<8, 7>), dim3(seg_blocks), dim3(512), seg_doubles * sizeof(double), ctx->stream, SDX_SEG_ARGS);
#undef SDX_SEG_ARGS
            }
            int rc = check_launch("k_raytrace_seg");
            if (rc) return rc;
            continue;
        }
        // the tolerance path (mixed_precision = 1): plane-parallel, all angles in one launch, flux only -> the fp32 recurrence
        if (ctx->mixed_precision && P == 1 && !inward && !acc && !inus && F && n_theta <= 64) {
            const int g32 = 64 / G;
            const size_t shmem32 = ((((size_t)(n_depth - 1) * nth + 2 * (size_t)n_depth + 3) & ~(size_t)3) +
                                    (size_t)(kRtBlock / 64) * (4 * (size_t)g32 * n_depth + kRt32Batch * (size_t)g32 * G)) * sizeof(float);
            if (shmem32 <= 64 * 1024) {
                {
                    LaunchScope ls(ctx, "k_raytrace", "k_raytrace_f32");
                    const unsigned blocks32 = (unsigned)((n_nu + (int64_t)g32 * (kRtBlock / 64) - 1) / ((int64_t)g32 * (kRtBlock / 64)));
                    hipLaunchKernelGGL(k_raytrace_f32, dim3(blocks32), dim3(kRtBlock), shmem32, ctx->stream, n_depth, n_nu, nth, n_theta, G, nus, temps, rd, w, alphas,
                                       ald, F, fld, g32, ft);
                }
                int rc = check_launch("k_raytrace_f32");
                if (rc) return rc;
                continue;
            }
        }
        {
            LaunchScope ls(ctx, "k_raytrace", shmem <= 64 * 1024 ? (P == 1 ? "k_raytrace<1>" : (P == 2 ? "k_raytrace<2>" : "k_raytrace<4>")) : "k_raytrace_basic");
#define SDX_RT_ARGS n_depth, n_nu, nth, n_theta, G, nus, temps, rd, w, alphas, ald, F, fld, inus, acc
            if (shmem <= 64 * 1024) {
                if (P == 1) hipLaunchKernelGGL(k_raytrace<1>, dim3(blocks), dim3(kRtBlock), shmem, ctx->stream, SDX_RT_ARGS, inward, gpw, ft);
                else if (P == 2) hipLaunchKernelGGL(k_raytrace<2>, dim3(blocks), dim3(kRtBlock), shmem, ctx->stream, SDX_RT_ARGS, inward, gpw, ft);
                else hipLaunchKernelGGL(k_raytrace<4>, dim3(blocks), dim3(kRtBlock), shmem, ctx->stream, SDX_RT_ARGS, inward, gpw, ft);
            } else {  // very deep models: the column does not fit LDS, recompute per lane instead
                REQUIRE(!ft.cont && !ft.source, "raytrace: fused total / caller's source plane not available for models this deep");
                REQUIRE(!inward, "raytrace: spherical geometry needs (3*n_depth*64/n_theta + 2*n_depth*n_theta) doubles of LDS per wave; model too deep");
                if (P == 1) hipLaunchKernelGGL(k_raytrace_basic<1>, dim3(blocks_basic), dim3(kBlock), 0, ctx->stream, SDX_RT_ARGS);
                else if (P == 2) hipLaunchKernelGGL(k_raytrace_basic<2>, dim3(blocks_basic), dim3(kBlock), 0, ctx->stream, SDX_RT_ARGS);
                else hipLaunchKernelGGL(k_raytrace_basic<4>, dim3(blocks_basic), dim3(kBlock), 0, ctx->stream, SDX_RT_ARGS);
            }
#undef SDX_RT_ARGS
        }
        int rc = check_launch("k_raytrace");
        if (rc) return rc;
    }
    return SDX_OK;
}

int sdx_raytrace_f64(sdx_ctx* ctx, int n_depth, int64_t n_nu, int n_theta, const double* nus, const double* temps,
                     const double* ray_dist, const double* wts, const double* alphas, double* F, double* I_nus)
{
    REQUIRE(ctx && n_depth >= 2 && n_nu >= 0 && n_theta > 0, "raytrace: need n_depth >= 2, n_theta > 0");
    REQUIRE(n_nu == 0 || (nus && temps && ray_dist && wts && alphas && F), "raytrace: null pointer");
    if (n_nu == 0) return SDX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc;
    const size_t f8 = sizeof(double), plane = (size_t)n_depth * n_nu * f8;
    const size_t in_bytes[] = {(size_t)n_nu * f8, (size_t)n_depth * f8, (size_t)(n_depth - 1) * n_theta * f8, (size_t)n_theta * f8, plane, plane};
    size_t dev_need = 256, pin_need = HostIo::pad(plane) + 256;
    for (size_t b : in_bytes) dev_need += HostIo::pad(b), pin_need += HostIo::pad(b);
    if (I_nus) dev_need += HostIo::pad(plane * n_theta), pin_need += HostIo::pad(plane * n_theta);
    HostIo io{ctx};
    if ((rc = io.begin(dev_need, pin_need))) return rc;
    const double *d_nus, *d_t, *d_rd, *d_w, *d_a, *d_f;
    if ((rc = io.upload(nus, in_bytes[0], (const void**)&d_nus)) || (rc = io.upload(temps, in_bytes[1], (const void**)&d_t)) ||
        (rc = io.upload(ray_dist, in_bytes[2], (const void**)&d_rd)) || (rc = io.upload(wts, in_bytes[3], (const void**)&d_w)) ||
        (rc = io.upload(alphas, in_bytes[4], (const void**)&d_a)) || (rc = io.upload(F, in_bytes[5], (const void**)&d_f)))  // F_nu is accumulated into (base.py:336)
        return rc;
    double* d_i = I_nus ? (double*)io.alloc(plane * n_theta) : nullptr;
    rc = sdx_raytrace_dev(ctx, n_depth, n_nu, n_theta, d_nus, d_t, d_rd, d_w, d_a, n_nu, (double*)d_f, n_nu, d_i, 1);
    if (rc) return rc;
    if ((rc = io.download(F, d_f, plane))) return rc;
    if (I_nus && (rc = io.download(I_nus, d_i, plane * n_theta))) return rc;
    return io.finish();
}

// ================================================================================================ post-processing
int sdx_convolve1d_reflect_dev(sdx_ctx* ctx, int64_t n, const double* in, int m, const double* weights, int symmetric, double* out)
{
    REQUIRE(ctx && n >= 0 && m > 0 && (m & 1), "convolve1d: need an odd kernel length");
    if (n == 0) return SDX_OK;
    REQUIRE(in && weights && out && in != out, "convolve1d: null or aliased pointer");
    {
        LaunchScope ls(ctx, "k_convolve1d_reflect");
        hipLaunchKernelGGL(k_convolve1d_reflect, dim3(blocks1(n)), dim3(kBlock), 0, ctx->stream, n, in, m, weights, symmetric, out);
    }
    return check_launch("k_convolve1d_reflect");
}

int sdx_flux_nu_to_lambda_dev(sdx_ctx* ctx, int64_t n, const double* f_nu, const double* nus, const double* lambdas, double* out)
{
    REQUIRE(ctx && n >= 0 && (n == 0 || (f_nu && nus && lambdas && out)), "flux_nu_to_lambda: bad arguments");
    if (n == 0) return SDX_OK;
    {
        LaunchScope ls(ctx, "k_flux_nu_to_lambda");
        hipLaunchKernelGGL(k_flux_nu_to_lambda, dim3(blocks1(n)), dim3(kBlock), 0, ctx->stream, n, f_nu, nus, lambdas, out);
    }
    return check_launch("k_flux_nu_to_lambda");
}

// ================================================================================================ fused synthesis
static int synthesize_impl(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t nu_begin, int64_t nu_count,
                           int64_t n_lines, const double* line_nus, const double* doppler, const double* gammas, int gamma_cols,
                           const double* alphas, const sdx_continuum* cont, int n_theta, const double* temps,
                           const double* ray_dist, const double* wts, double* alpha_line_out, double* total_alphas, double* F_nu,
                           int64_t ld, int64_t* n_evaluations_dev, const LineParams* gen, double* I_nus = nullptr, const double* source = nullptr,
                           int64_t source_ld = 0, const sdx_synthesis_options* opt = nullptr)
{
    int rc;
    REQUIRE(cont && temps && ray_dist && wts && n_theta > 0 && n_depth >= 2, "synthesize: bad arguments");
    REQUIRE(nu_begin >= 0 && nu_count >= 0 && nu_begin + nu_count <= n_nu, "synthesize: shard outside the grid");
    REQUIRE(nu_count == 0 || (F_nu && ld >= nu_count), "synthesize: bad output buffers");
    if ((rc = check_file_planes(cont, n_nu))) return rc;
    const int inward = opt && opt->inward_rays ? 1 : 0;
    const int n_extra = opt ? opt->n_line_planes : 0;
    REQUIRE(n_extra >= 0 && n_extra <= 2, "synthesize: n_line_planes must be 0..2");
    for (int k = 0; k < n_extra; ++k) REQUIRE(opt->line_plane[k] && opt->line_plane_ld >= nu_count, "synthesize: bad line plane");
    if (nu_count == 0) return SDX_OK;
    // Three launches on one stream: [pre-pass + continuum plane] -> [wide + narrow line kernels] -> [raytrace, which
    // forms total = continuum + line while staging its columns].  Independent work shares a launch instead of a
    // second stream: inter-queue edges cost ~12 us each on this platform, a whole kernel's worth at this size.
    int rc2 = ensure(ctx, &ctx->cont_ws, &ctx->cont_ws_bytes, (size_t)n_depth * nu_count * sizeof(double));
    if (rc2) return rc2;
    double* cont_plane = (double*)ctx->cont_ws;
    // the formal solution forms total = continuum + line planes while staging its columns when those fit LDS
    const size_t lds_columns = ((size_t)(kRtBlock / 64) * (2 * (size_t)n_depth + 4 * 64)) * sizeof(double);  // k_raytrace with one frequency per wave
    const bool fuse = n_theta <= 64 && lds_columns <= 64 * 1024;
    // spherical geometry (opt->inward_rays): the inward sweep before the outward one, then F_nu *= (r[-1] / reference_r)^2
    // (radiation_field_solvers/base.py:141-198, :340-344)
    auto finish = [&](int rc_trace) -> int {
        if (rc_trace || !inward) return rc_trace;
        {
            LaunchScope ls(ctx, "k_scale");
            hipLaunchKernelGGL(k_scale, grid2(nu_count, n_depth), dim3(kBlock), 0, ctx->stream, n_depth, nu_count, F_nu, ld, opt->photospheric_correction);
        }
        return check_launch("k_scale");
    };
    const ContinuumJob job{cont, nu_begin, nu_count, cont_plane};
    const double* part = nullptr;
    int64_t pld = 0;
    int n_planes = 0;
    // two-collective mode: the classification launch ran already (sdx_synthesize_classify_dev), the per-line maxima were gathered
    const ClassifyPhase second{2, 0, 0, opt ? const_cast<double*>(opt->line_m_max) : nullptr};
    REQUIRE(!second.m_max || (n_lines > 0 && !gen), "synthesize: options->line_m_max goes with a dense line list");
    if (!second.m_max) ctx->classified.valid = false;  // (this step rewrites the continuum plane a phase 1 left behind)
    if (n_lines > 0) {
        LineWork w;
        rc = line_partials(ctx, n_depth, n_nu, nus, nu_begin, nu_count, n_lines, line_nus, doppler, gammas, gamma_cols, alphas, &part, &pld,
                           &n_planes, &w, n_evaluations_dev != nullptr, &job, gen, second.m_max ? &second : nullptr);
        if (rc) return rc;
        if (n_evaluations_dev)
            HIP_TRY(hipMemcpyAsync(n_evaluations_dev, w.evals, sizeof(int64_t), hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        rc = launch_total(ctx, n_depth, nus, nu_begin, nu_count, cont, nullptr, 0, 1, nullptr, 0, cont_plane, nu_count);
        if (rc) return rc;
        if (n_evaluations_dev) HIP_TRY(hipMemsetAsync(n_evaluations_dev, 0, sizeof(int64_t), ctx->stream));  // (an empty list evaluates nothing)
        if (alpha_line_out)
            HIP_TRY(hipMemset2DAsync(alpha_line_out, ld * sizeof(double), 0, nu_count * sizeof(double), n_depth, ctx->stream));
    }
    if (!fuse) {
        // total = continuum (+ line planes), element-wise; without a caller buffer the continuum plane becomes the total in place
        double* total = total_alphas ? total_alphas : cont_plane;
        const int64_t tld = total_alphas ? ld : nu_count;
        {
            LaunchScope ls(ctx, "k_reduce_partials");
            if (total_alphas)
                HIP_TRY(hipMemcpy2DAsync(total_alphas, ld * sizeof(double), cont_plane, nu_count * sizeof(double), nu_count * sizeof(double),
                                         n_depth, hipMemcpyDeviceToDevice, ctx->stream));
            if (part) {
                if (alpha_line_out)
                    hipLaunchKernelGGL(k_reduce_partials, grid2(nu_count, n_depth), dim3(kBlock), 0, ctx->stream, n_depth, nu_count,
                                       n_planes, part, pld, alpha_line_out, ld, 0);
                hipLaunchKernelGGL(k_reduce_partials, grid2(nu_count, n_depth), dim3(kBlock), 0, ctx->stream, n_depth, nu_count, n_planes,
                                   part, pld, total, tld, 1);
            }
        }
        rc = check_launch("k_reduce_partials");
        if (rc) return rc;
        for (int k = 0; k < n_extra; ++k)
            if ((rc = sdx_accumulate_dev(ctx, n_depth, nu_count, total, tld, opt->line_plane[k], opt->line_plane_ld))) return rc;
        FusedTotal only_source{};  // (no fused total: the formal solution reads `total`; the caller's source plane still applies)
        only_source.source = source;
        only_source.sld = source_ld;
        return finish(raytrace_impl(ctx, n_depth, nu_count, n_theta, nus + nu_begin, temps, ray_dist, wts, total, tld, F_nu, ld, I_nus, 0, inward,
                                    source ? &only_source : nullptr, n_nu));
    }
    FusedTotal ft{};
    ft.source = source;
    ft.sld = source_ld;
    ft.cont = cont_plane;
    ft.cld = nu_count;
    ft.planes = part;
    ft.n_planes = n_planes;
    ft.pld = pld;
    ft.total_out = total_alphas;
    ft.line_out = part ? alpha_line_out : nullptr;
    ft.out_ld = ld;
    ft.n_extra = n_extra;
    for (int k = 0; k < n_extra; ++k) ft.extra[k] = opt->line_plane[k];
    ft.eld = n_extra ? opt->line_plane_ld : 0;
    return finish(raytrace_impl(ctx, n_depth, nu_count, n_theta, nus + nu_begin, temps, ray_dist, wts, nullptr, 0, F_nu, ld, I_nus, 0, inward, &ft, n_nu));
}

int sdx_synthesize_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t nu_begin, int64_t nu_count,
                       int64_t n_lines, const double* line_nus, const double* doppler, const double* gammas, int gamma_cols,
                       const double* alphas, const sdx_continuum* cont, int n_theta, const double* temps,
                       const double* ray_dist, const double* wts, double* alpha_line_out, double* total_alphas, double* F_nu,
                       int64_t ld, int64_t* n_evaluations_dev)
{
    int rc = check_line_args(ctx, n_depth, n_nu, nus, n_lines, line_nus, doppler, gammas, gamma_cols, alphas);
    if (rc) return rc;
    return synthesize_impl(ctx, n_depth, n_nu, nus, nu_begin, nu_count, n_lines, line_nus, doppler, gammas, gamma_cols, alphas, cont, n_theta,
                           temps, ray_dist, wts, alpha_line_out, total_alphas, F_nu, ld, n_evaluations_dev, nullptr);
}

int sdx_synthesize_ex_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t nu_begin, int64_t nu_count,
                          int64_t n_lines, const double* line_nus, const double* doppler, const double* gammas, int gamma_cols,
                          const double* alphas, const sdx_continuum* cont, int n_theta, const double* temps,
                          const double* ray_dist, const double* wts, double* alpha_line_out, double* total_alphas, double* F_nu,
                          int64_t ld, const double* source, int64_t source_ld, double* I_nus, int64_t* n_evaluations_dev)
{
    int rc = check_line_args(ctx, n_depth, n_nu, nus, n_lines, line_nus, doppler, gammas, gamma_cols, alphas);
    if (rc) return rc;
    REQUIRE(!source || source_ld >= nu_count, "synthesize_ex: source_ld must cover the columns");
    return synthesize_impl(ctx, n_depth, n_nu, nus, nu_begin, nu_count, n_lines, line_nus, doppler, gammas, gamma_cols, alphas, cont, n_theta,
                           temps, ray_dist, wts, alpha_line_out, total_alphas, F_nu, ld, n_evaluations_dev, nullptr, I_nus, source, source_ld);
}

int sdx_synthesize_linelist_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t nu_begin, int64_t nu_count,
                                const sdx_linelist* ll, const sdx_continuum* cont, int n_theta, const double* temps,
                                const double* ray_dist, const double* wts, double* alpha_line_out, double* total_alphas,
                                double* F_nu, int64_t ld, int64_t* n_evaluations_dev)
{
    REQUIRE(ctx, "null context");
    REQUIRE(n_depth > 0 && n_nu >= 0 && n_nu < (int64_t)2147483647, "synthesize: bad sizes");
    REQUIRE(n_nu == 0 || nus, "synthesize: null frequency grid");
    LineParams lp{};
    int rc = to_line_params(ll, n_depth, &lp);
    if (rc) return rc;
    return synthesize_impl(ctx, n_depth, n_nu, nus, nu_begin, nu_count, ll->n_lines, ll->nu, nullptr, nullptr,
                           ll->gamma_mode >= 2 ? 1 : n_depth, nullptr, cont, n_theta, temps, ray_dist, wts, alpha_line_out, total_alphas,
                           F_nu, ld, n_evaluations_dev, &lp);
}

int sdx_synthesize_opt_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t nu_begin, int64_t nu_count,
                           int64_t n_lines, const double* line_nus, const double* doppler, const double* gammas, int gamma_cols,
                           const double* alphas, const sdx_continuum* cont, int n_theta, const double* temps,
                           const double* ray_dist, const double* wts, double* alpha_line_out, double* total_alphas, double* F_nu,
                           int64_t ld, const sdx_synthesis_options* opt, int64_t* n_evaluations_dev)
{
    REQUIRE(ctx && opt, "synthesize_opt: null context or options");
    REQUIRE(!opt->source || opt->source_ld >= nu_count, "synthesize_opt: source_ld must cover the columns");
    if (opt->linelist) {  // line parameters generated in the pre-pass (f1): the dense arrays are not read
        REQUIRE(n_depth > 0 && n_nu >= 0 && n_nu < (int64_t)2147483647 && (n_nu == 0 || nus), "synthesize_opt: bad sizes");
        LineParams lp{};
        int rc = to_line_params(opt->linelist, n_depth, &lp);
        if (rc) return rc;
        return synthesize_impl(ctx, n_depth, n_nu, nus, nu_begin, nu_count, opt->linelist->n_lines, opt->linelist->nu, nullptr, nullptr,
                               opt->linelist->gamma_mode >= 2 ? 1 : n_depth, nullptr, cont, n_theta, temps, ray_dist, wts, alpha_line_out,
                               total_alphas, F_nu, ld, n_evaluations_dev, &lp, opt->I_nus, opt->source, opt->source_ld, opt);
    }
    int rc = check_line_args(ctx, n_depth, n_nu, nus, n_lines, line_nus, doppler, gammas, gamma_cols, alphas);
    if (rc) return rc;
    return synthesize_impl(ctx, n_depth, n_nu, nus, nu_begin, nu_count, n_lines, line_nus, doppler, gammas, gamma_cols, alphas, cont, n_theta,
                           temps, ray_dist, wts, alpha_line_out, total_alphas, F_nu, ld, n_evaluations_dev, nullptr, opt->I_nus, opt->source,
                           opt->source_ld, opt);
}

// Phase 1 of the two-collective mode (include/stardis_hip.h): the classification launch of a culled shard's step on a SHARE of the
// lines.  Everything else that launch does for the step — the grid-spacing partials, the shard's line ranges, the continuum plane —
// stays in the context's scratch for the synthesis that follows with options->line_m_max.
int sdx_synthesize_classify_dev(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t nu_begin, int64_t nu_count, int64_t n_lines,
                                const double* line_nus, const double* doppler, const double* gammas, int gamma_cols, const double* alphas,
                                const sdx_continuum* cont, int64_t line_begin, int64_t line_count, double* m_max)
{
    int rc = check_line_args(ctx, n_depth, n_nu, nus, n_lines, line_nus, doppler, gammas, gamma_cols, alphas);
    if (rc) return rc;
    REQUIRE(cont && m_max && n_lines > 0, "classify: null pointer or empty line list");
    REQUIRE(nu_begin >= 0 && nu_count > 0 && nu_begin + nu_count <= n_nu, "classify: shard outside the grid");
    REQUIRE(line_begin >= 0 && line_count >= 0 && line_begin + line_count <= n_lines, "classify: line share outside the list");
    if ((rc = check_file_planes(cont, n_nu))) return rc;
    if ((rc = ensure(ctx, &ctx->cont_ws, &ctx->cont_ws_bytes, (size_t)n_depth * nu_count * sizeof(double)))) return rc;
    // (the partial planes of the line kernels: reserved here so that the synthesis that follows moves nothing)
    if ((rc = ensure(ctx, &ctx->part_ws, &ctx->part_ws_bytes, (size_t)3 * n_depth * nu_count * sizeof(double)))) return rc;
    const ContinuumJob job{cont, nu_begin, nu_count, (double*)ctx->cont_ws};
    const ClassifyPhase first{1, line_begin, line_count, m_max};
    ctx->classified.valid = false;
    ctx->far_req_done = false;  // (request_far_ranges resets it only when the far field is on)
    if (far_field_on(ctx, n_nu) && (rc = request_far_ranges(ctx, n_nu, nu_begin, nu_count))) return rc;
    rc = line_prepass(ctx, n_depth, n_nu, nus, n_lines, line_nus, doppler, gammas, gamma_cols, alphas, true, nullptr, nullptr, nullptr, false, &job,
                      nullptr, nu_begin, nu_count, &first);
    ctx->far_req = FarReq{nullptr, 0, 0};
    return rc;
}

// The fused synthesis for a caller that holds everything in host memory (C, or numpy through ctypes): uploads, runs
// sdx_synthesize_dev, downloads.  `cont` carries HOST pointers here; array lengths follow from the sizes in the struct.
static int synthesize_host_check(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t n_lines, const double* line_nus,
                                 const double* doppler, const double* gammas, int gamma_cols, const double* alphas, const sdx_continuum* cont,
                                 int n_theta, const double* temps, const double* ray_dist, const double* wts)
{
    int rc = check_line_args(ctx, n_depth, n_nu, nus, n_lines, line_nus, doppler, gammas, gamma_cols, alphas);
    if (rc) return rc;
    REQUIRE(cont && temps && ray_dist && wts && n_theta > 0 && n_depth >= 2, "synthesize: bad arguments");
    if ((rc = host_grid_check(n_nu, nus))) return rc;
    if ((rc = host_lines_check(n_lines, line_nus))) return rc;
    for (int64_t k = 0; k < n_lines * n_depth; ++k)
        if (doppler[k] == 0.0) return fail(SDX_ERR_ARG, "doppler_width == 0 (ZeroDivisionError in the reference, voigt.py:148)");
    REQUIRE(cont->bf_n_species == 0 || !cont->bf_cutoff || cont->bf_n_levels > 0, "synthesize: bf_n_levels must be set");
    REQUIRE(cont->n_file_planes == 0, "synthesize: file planes are device planes (sdx_synthesize_dev); the host-buffer entry points take tabulated sources as the 1-D table only");
    return SDX_OK;
}

// Uploads everything to ctx's device, enqueues the synthesis of columns [nu_begin, nu_begin + nu_count) and the downloads of the
// requested planes into the caller's [n_depth][n_nu] host arrays (the shard's columns only).  `io` stays alive until the caller
// has called io.finish(); *d_F_out is the shard's device flux plane [n_depth][nu_count] (valid until the next call on ctx).
static int synthesize_host_shard(sdx_ctx* ctx, HostIo& io, int n_depth, int64_t n_nu, const double* nus, int64_t nu_begin, int64_t nu_count,
                                 int64_t n_lines, const double* line_nus, const double* doppler, const double* gammas, int gamma_cols,
                                 const double* alphas, const sdx_continuum* cont, int n_theta, const double* temps, const double* ray_dist,
                                 const double* wts, double* alpha_line_out, double* total_alphas, double* F_nu, int64_t* n_evaluations,
                                 double** d_F_out, size_t extra_dev_bytes = 0)
{
    int rc;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t f8 = sizeof(double), plane = (size_t)n_depth * nu_count * f8, ld = (size_t)n_lines * n_depth * f8;
    sdx_continuum c = *cont;
    const int n_levels = c.bf_n_species > 0 ? c.bf_n_levels : 0;
    // every non-null host array goes up; lengths follow from the sizes in the struct
    struct In {
        const void* src;
        size_t bytes;
        const void** dst;
    };
    const double *d_nus, *d_ln, *d_dw, *d_g, *d_a, *d_t, *d_rd, *d_w;
    const In ins[] = {
        {nus, (size_t)n_nu * f8, (const void**)&d_nus},
        {line_nus, (size_t)n_lines * f8, (const void**)&d_ln},
        {doppler, ld, (const void**)&d_dw},
        {gammas, (size_t)n_lines * gamma_cols * f8, (const void**)&d_g},
        {alphas, ld, (const void**)&d_a},
        {temps, (size_t)n_depth * f8, (const void**)&d_t},
        {ray_dist, (size_t)(n_depth - 1) * n_theta * f8, (const void**)&d_rd},
        {wts, (size_t)n_theta * f8, (const void**)&d_w},
        {cont->lambdas, (size_t)n_nu * f8, (const void**)&c.lambdas},
        {cont->table_wavelength, (size_t)c.n_table * f8, (const void**)&c.table_wavelength},
        {cont->table_sigma, (size_t)c.n_table * f8, (const void**)&c.table_sigma},
        {cont->table_density, (size_t)n_depth * f8, (const void**)&c.table_density},
        {cont->bf_species_offsets, (size_t)(c.bf_n_species + 1) * sizeof(int32_t), (const void**)&c.bf_species_offsets},
        {cont->bf_species_ion_number, (size_t)c.bf_n_species * sizeof(int32_t), (const void**)&c.bf_species_ion_number},
        {cont->bf_cutoff, (size_t)n_levels * f8, (const void**)&c.bf_cutoff},
        {cont->bf_level_density, (size_t)n_levels * n_depth * f8, (const void**)&c.bf_level_density},
        {cont->ff_species_ion_number, (size_t)c.ff_n_species * sizeof(int32_t), (const void**)&c.ff_species_ion_number},
        {cont->ff_number_density, (size_t)c.ff_n_species * n_depth * f8, (const void**)&c.ff_number_density},
        {cont->ray_n_h, (size_t)n_depth * f8, (const void**)&c.ray_n_h},
        {cont->ray_n_he, (size_t)n_depth * f8, (const void**)&c.ray_n_he},
        {cont->ray_n_h2, (size_t)n_depth * f8, (const void**)&c.ray_n_h2},
        {cont->electron_density, (size_t)n_depth * f8, (const void**)&c.electron_density},
    };
    const int n_out = 1 + (alpha_line_out ? 1 : 0) + (total_alphas ? 1 : 0);
    size_t dev_need = (size_t)n_out * HostIo::pad(plane) + 512 + HostIo::pad(extra_dev_bytes);
    size_t pin_need = (size_t)(F_nu ? n_out : n_out - 1) * HostIo::pad(plane) + 512 + HostIo::pad(extra_dev_bytes);
    for (const In& in : ins)
        if (in.src) dev_need += HostIo::pad(in.bytes ? in.bytes : 8), pin_need += HostIo::pad(in.bytes);
    if ((rc = io.begin(dev_need, pin_need))) return rc;
    for (const In& in : ins)
        if (in.src && (rc = io.upload(in.src, in.bytes, in.dst))) return rc;
    c.temperature = d_t;
    double* d_line = alpha_line_out ? (double*)io.alloc(plane) : nullptr;
    double* d_total = total_alphas ? (double*)io.alloc(plane) : nullptr;
    double* d_F = (double*)io.alloc(plane);
    int64_t* d_ev = (int64_t*)io.alloc(sizeof(int64_t));
    rc = sdx_synthesize_dev(ctx, n_depth, n_nu, d_nus, nu_begin, nu_count, n_lines, d_ln, d_dw, d_g, gamma_cols, d_a, &c, n_theta, d_t, d_rd, d_w,
                            d_line, d_total, d_F, nu_count, n_evaluations ? d_ev : nullptr);
    if (rc) return rc;
    const size_t pitch = (size_t)n_nu * f8, row = (size_t)nu_count * f8;
    if (alpha_line_out && (rc = io.download2d(alpha_line_out + nu_begin, pitch, d_line, row, n_depth))) return rc;
    if (total_alphas && (rc = io.download2d(total_alphas + nu_begin, pitch, d_total, row, n_depth))) return rc;
    if (F_nu && (rc = io.download2d(F_nu + nu_begin, pitch, d_F, row, n_depth))) return rc;
    if (n_evaluations) HIP_TRY(hipMemcpyAsync(n_evaluations, d_ev, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    if (d_F_out) *d_F_out = d_F;
    return SDX_OK;
}

int sdx_synthesize_f64(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t n_lines, const double* line_nus,
                       const double* doppler, const double* gammas, int gamma_cols, const double* alphas, const sdx_continuum* cont,
                       int n_theta, const double* temps, const double* ray_dist, const double* wts, double* alpha_line_out,
                       double* total_alphas, double* F_nu, int64_t* n_evaluations)
{
    int rc = synthesize_host_check(ctx, n_depth, n_nu, nus, n_lines, line_nus, doppler, gammas, gamma_cols, alphas, cont, n_theta, temps, ray_dist, wts);
    if (rc) return rc;
    REQUIRE(n_nu == 0 || F_nu, "synthesize: null output");
    if (n_nu == 0) return SDX_OK;
    HostIo io{ctx};
    int64_t ev = 0;
    rc = synthesize_host_shard(ctx, io, n_depth, n_nu, nus, 0, n_nu, n_lines, line_nus, doppler, gammas, gamma_cols, alphas, cont, n_theta, temps,
                               ray_dist, wts, alpha_line_out, total_alphas, F_nu, n_evaluations ? &ev : nullptr, nullptr);
    if (rc) return rc;
    if ((rc = io.finish())) return rc;
    if (n_evaluations) *n_evaluations = ev;
    return SDX_OK;
}

// The continuum sources for a caller that holds everything in host memory: what calc_alpha_file / _bf / _ff / _rayleigh /
// _electron (opacities_solvers/base.py:40-317) return, and their sum in calc_alphas' order (:655-700), from ONE upload of the
// per-depth vectors and the table — on the context's persistent staging like the other *_f64 entry points.  Each plane is
// produced by the device function the per-source *_dev entry point runs (the same bits); `total` by k_total_alphas.
int sdx_continuum_f64(sdx_ctx* ctx, int n_depth, int64_t n_nu, double* nus, const sdx_continuum* cont, double* alpha_file,
                      double* alpha_bf, double* alpha_ff, double* alpha_rayleigh, double* alpha_electron, double* total_alphas)
{
    REQUIRE(ctx && cont && n_depth > 0 && n_nu >= 0, "continuum: bad arguments");
    REQUIRE(alpha_file || alpha_bf || alpha_ff || alpha_rayleigh || alpha_electron || total_alphas, "continuum: no output requested");
    if (n_nu == 0) return SDX_OK;
    REQUIRE(nus, "continuum: null frequency grid");
    REQUIRE(cont->n_file_planes == 0, "continuum: file planes are device planes; the host-buffer entry point takes the 1-D table only");
    const bool has_table = cont->table_sigma != nullptr, has_bf = cont->bf_cutoff && cont->bf_n_species > 0;
    const bool has_ff = cont->ff_number_density && cont->ff_n_species > 0;
    REQUIRE(!has_table || (cont->lambdas && cont->table_wavelength && cont->table_density && cont->n_table > 0), "continuum: incomplete table source");
    REQUIRE(!has_bf || (cont->bf_n_levels > 0 && cont->bf_n_levels <= 4096 && cont->bf_species_offsets && cont->bf_species_ion_number && cont->bf_level_density),
            "continuum: incomplete bound-free description (bf_n_levels must be set, 1..4096)");
    REQUIRE(!has_ff || (cont->ff_species_ion_number && cont->temperature), "continuum: incomplete free-free description");
    REQUIRE(!alpha_file || has_table, "continuum: alpha_file requested without a table");
    HIP_TRY(hipSetDevice(ctx->device));
    int rc;
    const size_t f8 = sizeof(double), plane = (size_t)n_depth * n_nu * f8;
    sdx_continuum c = *cont;
    const int n_levels = has_bf ? c.bf_n_levels : 0;
    struct In {
        const void* src;
        size_t bytes;
        const void** dst;
    };
    const double* d_nus;
    const In ins[] = {
        {nus, (size_t)n_nu * f8, (const void**)&d_nus},
        {cont->lambdas, (size_t)n_nu * f8, (const void**)&c.lambdas},
        {cont->table_wavelength, (size_t)c.n_table * f8, (const void**)&c.table_wavelength},
        {cont->table_sigma, (size_t)c.n_table * f8, (const void**)&c.table_sigma},
        {cont->table_density, (size_t)n_depth * f8, (const void**)&c.table_density},
        {cont->bf_species_offsets, (size_t)(c.bf_n_species + 1) * sizeof(int32_t), (const void**)&c.bf_species_offsets},
        {cont->bf_species_ion_number, (size_t)c.bf_n_species * sizeof(int32_t), (const void**)&c.bf_species_ion_number},
        {cont->bf_cutoff, (size_t)n_levels * f8, (const void**)&c.bf_cutoff},
        {cont->bf_level_density, (size_t)n_levels * n_depth * f8, (const void**)&c.bf_level_density},
        {cont->ff_species_ion_number, (size_t)c.ff_n_species * sizeof(int32_t), (const void**)&c.ff_species_ion_number},
        {cont->ff_number_density, (size_t)c.ff_n_species * n_depth * f8, (const void**)&c.ff_number_density},
        {cont->ray_n_h, (size_t)n_depth * f8, (const void**)&c.ray_n_h},
        {cont->ray_n_he, (size_t)n_depth * f8, (const void**)&c.ray_n_he},
        {cont->ray_n_h2, (size_t)n_depth * f8, (const void**)&c.ray_n_h2},
        {cont->electron_density, (size_t)n_depth * f8, (const void**)&c.electron_density},
        {cont->temperature, (size_t)n_depth * f8, (const void**)&c.temperature},
    };
    double* const outs[] = {total_alphas, alpha_file, alpha_bf, alpha_ff, alpha_rayleigh, alpha_electron};
    int n_out = 0;
    for (double* o : outs) n_out += o ? 1 : 0;
    size_t dev_need = (size_t)n_out * HostIo::pad(plane) + 512, pin_need = (size_t)n_out * HostIo::pad(plane) + HostIo::pad((size_t)n_nu * f8) + 512;
    for (const In& in : ins)
        if (in.src) dev_need += HostIo::pad(in.bytes ? in.bytes : 8), pin_need += HostIo::pad(in.bytes);
    HostIo io{ctx};
    if ((rc = io.begin(dev_need, pin_need))) return rc;
    for (const In& in : ins)
        if (in.src && (rc = io.upload(in.src, in.bytes, in.dst))) return rc;
    double* d_out[6];
    for (int k = 0; k < 6; ++k) d_out[k] = outs[k] ? (double*)io.alloc(plane) : nullptr;
    // the total first: it reads the caller's frequencies as they came (the Rayleigh source clips its own copy of a frequency
    // above 2.3e15 Hz, :99, whether or not the array has been clipped yet)
    if (total_alphas && (rc = launch_total(ctx, n_depth, d_nus, 0, n_nu, &c, nullptr, 0, 1, nullptr, 0, d_out[0], n_nu))) return rc;
    if (alpha_file && (rc = sdx_alpha_file_1d_dev(ctx, n_depth, n_nu, c.lambdas, c.n_table, c.table_wavelength, c.table_sigma, c.table_density, d_out[1], n_nu)))
        return rc;
    if (alpha_bf) {
        ContinuumArgs a{};
        if (has_bf) {
            double* coef = nullptr;
            if ((rc = launch_bf_coef(ctx, n_depth, c.bf_n_species, n_levels, c.bf_species_offsets, c.bf_species_ion_number, c.bf_cutoff, c.bf_level_density, &coef)))
                return rc;
            a.bf_n_species = c.bf_n_species;
            a.bf_species_offsets = (const int*)c.bf_species_offsets;
            a.bf_species_ion_number = (const int*)c.bf_species_ion_number;
            a.bf_cutoff = c.bf_cutoff;
            a.bf_coef = coef;
        }
        if ((rc = launch_source(ctx, "k_alpha_bf", kSrcBf, n_depth, n_nu, d_nus, a, d_out[2], n_nu))) return rc;
    }
    if (alpha_ff && (rc = sdx_alpha_ff_dev(ctx, n_depth, n_nu, d_nus, c.temperature ? c.temperature : d_nus, has_ff ? c.ff_n_species : 0, c.ff_species_ion_number,
                                           c.ff_number_density, d_out[3], n_nu)))
        return rc;
    const bool clip = alpha_rayleigh != nullptr;  // calc_alpha_rayleigh zeroes the caller's frequencies above 2.3e15 Hz in place (:99)
    if (alpha_rayleigh && (rc = sdx_alpha_rayleigh_dev(ctx, n_depth, n_nu, (double*)d_nus, c.ray_n_h, c.ray_n_he, c.ray_n_h2, d_out[4], n_nu))) return rc;
    if (alpha_electron) {
        if (c.electron_density) {
            if ((rc = sdx_alpha_electron_dev(ctx, n_depth, n_nu, c.electron_density, d_out[5], n_nu))) return rc;
        } else {
            HIP_TRY(hipMemsetAsync(d_out[5], 0, plane, ctx->stream));  // (the reference returns the scalar 0 when disabled, :164-165)
        }
    }
    for (int k = 0; k < 6; ++k)
        if (outs[k] && (rc = io.download(outs[k], d_out[k], plane))) return rc;
    if (clip && (rc = io.download(nus, d_nus, (size_t)n_nu * f8))) return rc;
    return io.finish();
}

// ---- fp32-mixed twins of the host-buffer entry points (include/stardis_hip.h): the option on for the call, restored afterwards
namespace {
struct MixedScope {
    sdx_ctx* ctx;
    int64_t saved;
    explicit MixedScope(sdx_ctx* c) : ctx(c), saved(c ? c->mixed_precision : 0)
    {
        if (ctx) ctx->mixed_precision = 1;
    }
    ~MixedScope()
    {
        if (ctx) ctx->mixed_precision = saved;
    }
};
}  // namespace

int sdx_line_opacity_f32mix(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t n_lines, const double* line_nus,
                            const double* doppler, const double* gammas, int gamma_cols, const double* alphas, double* out, int64_t* n_evaluations)
{
    REQUIRE(ctx, "null context");
    MixedScope on(ctx);
    return sdx_line_opacity_f64(ctx, n_depth, n_nu, nus, n_lines, line_nus, doppler, gammas, gamma_cols, alphas, out, n_evaluations);
}

int sdx_raytrace_f32mix(sdx_ctx* ctx, int n_depth, int64_t n_nu, int n_theta, const double* nus, const double* temps, const double* ray_dist,
                        const double* wts, const double* alphas, double* F, double* I_nus)
{
    REQUIRE(ctx, "null context");
    MixedScope on(ctx);
    return sdx_raytrace_f64(ctx, n_depth, n_nu, n_theta, nus, temps, ray_dist, wts, alphas, F, I_nus);
}

int sdx_synthesize_f32mix(sdx_ctx* ctx, int n_depth, int64_t n_nu, const double* nus, int64_t n_lines, const double* line_nus, const double* doppler,
                          const double* gammas, int gamma_cols, const double* alphas, const sdx_continuum* cont, int n_theta, const double* temps,
                          const double* ray_dist, const double* wts, double* alpha_line_out, double* total_alphas, double* F_nu, int64_t* n_evaluations)
{
    REQUIRE(ctx, "null context");
    MixedScope on(ctx);
    return sdx_synthesize_f64(ctx, n_depth, n_nu, nus, n_lines, line_nus, doppler, gammas, gamma_cols, alphas, cont, n_theta, temps, ray_dist, wts,
                              alpha_line_out, total_alphas, F_nu, n_evaluations);
}

}  // extern "C"

// ================================================================================================ one process, several GPUs
// the frequency axis shards with no data-path exchange (radiation_field_solvers/base.py:200 is a prange over
// nu); one process drives one context + stream per device, the line list is replicated, every device keeps the GLOBAL window
// rule, and ONE ncclAllGather (RCCL over xGMI) of the padded F_nu[-1] shards leaves the whole emergent spectrum on every device.
//
// RCCL is opened at run time (dlopen of librccl.so.1, RTLD_LOCAL), not linked: a process that has imported torch already holds
// torch's own copy under the same SONAME and the loader hands that one back — two RCCL builds never meet in one process — and a
// single-GPU user of this library does not map RCCL's half gigabyte of device code.  Every RCCL failure (library missing,
// communicator set-up, the collective itself) is SDX_ERR_COMM.
namespace {
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string error;
};

RcclApi* rccl_api()
{
    static std::mutex m;
    static RcclApi api;
    std::lock_guard<std::mutex> lock(m);
    if (api.handle) return &api;
    // Which RCCL: the one that sits NEXT TO THE HIP RUNTIME THIS LIBRARY IS BOUND TO.  A process may hold two ROCm stacks — the
    // system's and the one a PyTorch wheel bundles (its own libamdhip64 and librccl under torch/lib) — and whichever HIP runtime
    // was loaded first serves everybody; an RCCL from the other stack on top of it fails in ncclCommInitAll ("unhandled cuda
    // error": seen when this library was loaded before torch and "librccl.so.1" then resolved to torch's copy).  An absolute
    // path also keeps the loader from handing back a same-named library that is already mapped.  SDX_RCCL_LIB overrides.
    const char* env = std::getenv("SDX_RCCL_LIB");
    void* h = nullptr;
    std::string tried;
    if (env && *env) {
        h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
        tried = env;
    } else {
        Dl_info info{};
        std::string dir;
        if (dladdr((const void*)&hipGetDeviceCount, &info) && info.dli_fname) {
            dir = info.dli_fname;
            const size_t slash = dir.rfind('/');
            dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
        }
        const std::string candidates[] = {dir + "librccl.so.1", dir + "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const std::string& c : candidates) {
            if (c.empty() || (c[0] != '/' && c != "librccl.so.1")) continue;
            h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
            tried += (tried.empty() ? "" : ", ") + c;
            if (h) break;
        }
    }
    if (!h) {
        const char* e = dlerror();
        api.error = std::string("RCCL not available (tried ") + tried + "): " + (e ? e : "dlopen failed");
        return &api;
    }
    bool ok = true;
    auto sym = [&](const char* name) {
        void* p = dlsym(h, name);
        if (!p) ok = false, api.error = std::string("RCCL symbol missing: ") + name;
        return p;
    };
    api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    api.GetVersion = (decltype(api.GetVersion))sym("ncclGetVersion");
    if (!ok) {
        dlclose(h);
        return &api;
    }
    api.handle = h;
    api.error.clear();
    return &api;
}
}  // namespace

struct sdx_group {
    int n = 0;
    std::vector<int> devices;
    std::vector<sdx_ctx*> ctx;
    std::vector<ncclComm_t> comm;  // empty: loop-back test mode (SDX_GROUP_LOOPBACK=1, duplicate devices allowed, no RCCL)
    std::vector<double*> send, recv;  // per device: [per], [per * n]
    size_t per_cap = 0;
    // what the last sdx_synthesize_sharded_f64 did (sdx_group_last_gather)
    int64_t last_bytes_per_rank = 0;
    int last_ranks = 0;
};

#define NCCL_TRY(api, expr)                                                                                     \
    do {                                                                                                        \
        ncclResult_t r_ = (expr);                                                                               \
        if (r_ != ncclSuccess) return fail(SDX_ERR_COMM, std::string(#expr) + ": " + (api)->GetErrorString(r_)); \
    } while (0)

static int group_buffers(sdx_group* g, size_t per)
{
    if (g->per_cap >= per) return SDX_OK;
    for (int r = 0; r < g->n; ++r) {
        HIP_TRY(hipSetDevice(g->devices[r]));
        HIP_TRY(hipStreamSynchronize(g->ctx[r]->stream));
        if (g->send[r]) HIP_TRY(hipFree(g->send[r]));
        if (g->recv[r]) HIP_TRY(hipFree(g->recv[r]));
        g->send[r] = g->recv[r] = nullptr;
        HIP_TRY(hipMalloc((void**)&g->send[r], per * sizeof(double)));
        HIP_TRY(hipMalloc((void**)&g->recv[r], per * g->n * sizeof(double)));
    }
    g->per_cap = per;
    return SDX_OK;
}

extern "C" {

void sdx_group_destroy(sdx_group* g)
{
    if (!g) return;
    RcclApi* api = g->comm.empty() ? nullptr : rccl_api();
    for (int r = 0; r < (int)g->ctx.size(); ++r) {
        if (!g->ctx[r]) continue;
        hipSetDevice(g->devices[r]);
        hipStreamSynchronize(g->ctx[r]->stream);
        if (r < (int)g->send.size() && g->send[r]) hipFree(g->send[r]);
        if (r < (int)g->recv.size() && g->recv[r]) hipFree(g->recv[r]);
    }
    if (api && api->handle)
        for (ncclComm_t c : g->comm)
            if (c) api->CommDestroy(c);
    for (sdx_ctx* c : g->ctx) sdx_destroy(c);
    delete g;
}

sdx_group* sdx_group_create(int n_gpus, const int* devices)
{
    if (n_gpus <= 0 || n_gpus > 64) {
        fail(SDX_ERR_ARG, "sdx_group_create: n_gpus must be 1..64");
        return nullptr;
    }
    const bool loopback = knob("SDX_GROUP_LOOPBACK") && std::atoi(knob("SDX_GROUP_LOOPBACK")) == 1;  // test hook, see header
    // RCCL first: without it there is no group, whatever the devices
    RcclApi* api = loopback ? nullptr : rccl_api();
    if (api && !api->handle) {
        fail(SDX_ERR_COMM, api->error);
        return nullptr;
    }
    const int visible = sdx_device_count();
    std::unique_ptr<sdx_group> g(new sdx_group());
    g->n = n_gpus;
    for (int r = 0; r < n_gpus; ++r) {
        const int dev = devices ? devices[r] : r;
        if (dev < 0 || dev >= visible) {
            fail(SDX_ERR_ARG, "sdx_group_create: device " + std::to_string(dev) + " is not visible (sdx_device_count() = " + std::to_string(visible) + ")");
            return nullptr;
        }
        if (!loopback)
            for (int q = 0; q < r; ++q)
                if (g->devices[q] == dev) {
                    fail(SDX_ERR_ARG, "sdx_group_create: device " + std::to_string(dev) + " listed twice (one RCCL rank per GPU)");
                    return nullptr;
                }
        g->devices.push_back(dev);
    }
    g->send.assign(n_gpus, nullptr);
    g->recv.assign(n_gpus, nullptr);
    for (int r = 0; r < n_gpus; ++r) {
        sdx_ctx* c = sdx_create(g->devices[r], nullptr);
        if (!c) {
            sdx_group_destroy(g.release());
            return nullptr;
        }
        g->ctx.push_back(c);
    }
    if (!loopback) {
        g->comm.assign(n_gpus, nullptr);
        const ncclResult_t r = api->CommInitAll(g->comm.data(), n_gpus, g->devices.data());
        if (r != ncclSuccess) {
            const std::string msg = std::string("ncclCommInitAll: ") + api->GetErrorString(r);
            for (auto& c : g->comm) c = nullptr;
            sdx_group_destroy(g.release());
            fail(SDX_ERR_COMM, msg);
            return nullptr;
        }
    }
    return g.release();
}

int sdx_group_size(const sdx_group* g) { return g ? g->n : 0; }
sdx_ctx* sdx_group_context(sdx_group* g, int rank) { return g && rank >= 0 && rank < g->n ? g->ctx[rank] : nullptr; }

int sdx_group_last_gather(const sdx_group* g, int* ranks, int64_t* bytes_per_rank, int* rccl_version)
{
    REQUIRE(g, "null group");
    if (ranks) *ranks = g->last_ranks;
    if (bytes_per_rank) *bytes_per_rank = g->last_bytes_per_rank;
    if (rccl_version) {
        *rccl_version = 0;
        if (!g->comm.empty()) {
            RcclApi* api = rccl_api();
            if (api->handle) api->GetVersion(rccl_version);
        }
    }
    return SDX_OK;
}

int sdx_synthesize_sharded_f64(sdx_group* g, int n_depth, int64_t n_nu, const double* nus, int64_t n_lines, const double* line_nus,
                               const double* doppler, const double* gammas, int gamma_cols, const double* alphas, const sdx_continuum* cont,
                               int n_theta, const double* temps, const double* ray_dist, const double* wts, const int64_t* shard_begin,
                               double* alpha_line_out, double* total_alphas, double* F_nu, double* emergent_flux, int64_t* n_evaluations)
{
    REQUIRE(g, "null group");
    const int P = g->n;
    int rc = synthesize_host_check(g->ctx[0], n_depth, n_nu, nus, n_lines, line_nus, doppler, gammas, gamma_cols, alphas, cont, n_theta, temps, ray_dist, wts);
    if (rc) return rc;
    REQUIRE(n_nu == 0 || F_nu || emergent_flux, "synthesize_sharded: no output requested");
    if (n_nu == 0) return SDX_OK;
    // shards: contiguous blocks of the GLOBAL frequency index, equal by default
    std::vector<int64_t> begin(P + 1);
    const int64_t per_equal = (n_nu + P - 1) / P;
    for (int r = 0; r <= P; ++r) begin[r] = shard_begin ? shard_begin[r] : std::min<int64_t>((int64_t)r * per_equal, n_nu);
    REQUIRE(begin[0] == 0 && begin[P] == n_nu, "synthesize_sharded: shard_begin must run from 0 to n_nu");
    int64_t per = 0;
    for (int r = 0; r < P; ++r) {
        REQUIRE(begin[r + 1] >= begin[r], "synthesize_sharded: shard_begin must not descend");
        per = std::max(per, begin[r + 1] - begin[r]);
    }
    if ((rc = group_buffers(g, (size_t)per))) return rc;

    // per device, in parallel host threads (the staging memcpy of a replicated 10^6-line list is the slow part): upload,
    // enqueue the shard's synthesis, pack its F_nu[-1] columns into the (zero-padded) send buffer, enqueue the plane downloads
    std::vector<std::unique_ptr<HostIo>> io(P);
    std::vector<int> rcs(P, SDX_OK), codes(P, 0);
    std::vector<std::string> errs(P);
    int64_t ev = 0;
    // the evaluation count (every window of every line: a global figure) is taken from ONE rank — the first that has columns,
    // not blindly rank 0, whose shard may be empty; asking for it switches that rank's culled pre-pass off (it has to see every
    // window), which makes it the straggler of the group: a diagnostic, not something to request in a timed loop
    int ev_rank = -1;
    for (int r = 0; r < P && ev_rank < 0; ++r)
        if (begin[r + 1] > begin[r]) ev_rank = r;
    auto work = [&](int r) {
        sdx_ctx* ctx = g->ctx[r];
        io[r].reset(new HostIo(ctx));
        const int64_t b = begin[r], cnt = begin[r + 1] - begin[r];
        double* d_F = nullptr;
        int rr = SDX_OK;
        if (hipSetDevice(ctx->device) != hipSuccess) rr = fail(SDX_ERR_HIP, "hipSetDevice failed");
        if (!rr && cnt > 0)
            rr = synthesize_host_shard(ctx, *io[r], n_depth, n_nu, nus, b, cnt, n_lines, line_nus, doppler, gammas, gamma_cols, alphas, cont, n_theta,
                                       temps, ray_dist, wts, alpha_line_out, total_alphas, F_nu, (r == ev_rank && n_evaluations) ? &ev : nullptr, &d_F);
        if (!rr && hipMemsetAsync(g->send[r], 0, (size_t)per * sizeof(double), ctx->stream) != hipSuccess) rr = fail(SDX_ERR_HIP, "hipMemsetAsync(send) failed");
        if (!rr && cnt > 0 &&
            hipMemcpyAsync(g->send[r], d_F + (size_t)(n_depth - 1) * cnt, (size_t)cnt * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
            rr = fail(SDX_ERR_HIP, "hipMemcpyAsync(pack flux shard) failed");
        rcs[r] = rr;
        if (rr) errs[r] = g_error, codes[r] = g_error_code;
    };
    if (P == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int r = 0; r < P; ++r) th.emplace_back(work, r);
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < P; ++r)
        if (rcs[r]) return fail(rcs[r], "device " + std::to_string(g->devices[r]) + ": " + errs[r]);

    // the one collective: all-gather of the padded emergent-flux shards, stream-ordered behind each device's kernels
    if (!g->comm.empty()) {
        RcclApi* api = rccl_api();
        NCCL_TRY(api, api->GroupStart());
        for (int r = 0; r < P; ++r) {
            const ncclResult_t res = api->AllGather(g->send[r], g->recv[r], (size_t)per, ncclDouble, g->comm[r], g->ctx[r]->stream);
            if (res != ncclSuccess) {
                api->GroupEnd();
                return fail(SDX_ERR_COMM, std::string("ncclAllGather: ") + api->GetErrorString(res));
            }
        }
        NCCL_TRY(api, api->GroupEnd());
    } else {  // loop-back test mode: the same data movement as plain copies after a full synchronisation
        for (int r = 0; r < P; ++r) {
            HIP_TRY(hipSetDevice(g->devices[r]));
            HIP_TRY(hipStreamSynchronize(g->ctx[r]->stream));
        }
        for (int r = 0; r < P; ++r) {
            HIP_TRY(hipSetDevice(g->devices[r]));
            for (int q = 0; q < P; ++q)
                HIP_TRY(hipMemcpyAsync(g->recv[r] + (size_t)q * per, g->send[q], (size_t)per * sizeof(double), hipMemcpyDeviceToDevice, g->ctx[r]->stream));
        }
    }
    g->last_ranks = P;
    g->last_bytes_per_rank = per * (int64_t)sizeof(double);

    // the gathered spectrum comes back from device 0 (every device holds it), padding trimmed
    HIP_TRY(hipSetDevice(g->devices[0]));
    std::vector<double> gathered;
    if (emergent_flux) {
        gathered.resize((size_t)per * P);
        HIP_TRY(hipMemcpyAsync(gathered.data(), g->recv[0], gathered.size() * sizeof(double), hipMemcpyDeviceToHost, g->ctx[0]->stream));
    }
    for (int r = 0; r < P; ++r) {
        HIP_TRY(hipSetDevice(g->devices[r]));
        if ((rc = io[r]->finish())) return rc;
    }
    if (emergent_flux)
        for (int r = 0; r < P; ++r) std::memcpy(emergent_flux + begin[r], gathered.data() + (size_t)r * per, (size_t)(begin[r + 1] - begin[r]) * sizeof(double));
    if (n_evaluations) *n_evaluations = ev;
    return SDX_OK;
}

}  // extern "C"

#ifdef SDX_PRE_STATS
// analysis build only (scripts/r5/pre_stats.sh): the phase time stamps of the pre-pass blocks of the launches so far, then cleared
extern "C" int sdx_pre_stats_read(unsigned long long* out, long long n_words)
{
    const size_t bytes = std::min<size_t>((size_t)n_words * 8, sizeof(unsigned long long) * (size_t)sdx::kPreStatSlots * 8);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sdx::g_pre_stats), bytes) != hipSuccess) return -1;
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(sdx::g_pre_stats)) != hipSuccess) return -1;
    return hipMemset(p, 0, sizeof(unsigned long long) * (size_t)sdx::kPreStatSlots * 8) == hipSuccess ? 0 : -1;
}
#endif

#ifdef SDX_WALK_STATS
// analysis build only (scripts/r4/walk_stats.sh): the line kernel's wave statistics of the launches so far, then cleared
extern "C" int sdx_walk_stats_read(unsigned long long* out, long long n_words)
{
    const size_t bytes = std::min<size_t>((size_t)n_words * 8, sizeof(unsigned long long) * (size_t)sdx::kWalkStatSlots * 8);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sdx::g_walk_stats), bytes) != hipSuccess) return -1;
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(sdx::g_walk_stats)) != hipSuccess) return -1;
    return hipMemset(p, 0, sizeof(unsigned long long) * (size_t)sdx::kWalkStatSlots * 8) == hipSuccess ? 0 : -1;
}
#endif

