// Issue cost of the instructions the hot kernels are made of, on gfx950: cycles per wave-instruction for one wave's stream
// of INDEPENDENT instructions (8 chains) and of DEPENDENT ones (1 chain), at 1, 2, 4 and 8 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o scripts/_build/issue_cost scripts/issue_cost.hip && scripts/_build/issue_cost
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// body: asm statement using register a[k] (double) as source and destination
#define DEFINE_D(NAME, ASM)                                                                                   \
    __global__ void k_##NAME(unsigned long long* cyc, double* sink, int iters, int dep)                         \
    {                                                                                                         \
        double a[8];                                                                                          \
        for (int k = 0; k < 8; ++k) a[k] = 1.0 + 1e-3 * (threadIdx.x + k);                                      \
        const double c = 1.0000001, d = 1e-9;                                                                 \
        (void)c; (void)d;                                                                                     \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                           \
        if (dep) {                                                                                            \
            for (int i = 0; i < iters; ++i) {                                                                 \
                _Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile(ASM : "+v"(a[0]) : "v"(c), "v"(d)); \
            }                                                                                                 \
        } else {                                                                                              \
            for (int i = 0; i < iters; ++i) {                                                                 \
                _Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile(ASM : "+v"(a[k]) : "v"(c), "v"(d)); \
            }                                                                                                 \
        }                                                                                                     \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                           \
        double s = 0;                                                                                         \
        for (int k = 0; k < 8; ++k) s += a[k];                                                                \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                      \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;             \
    }

#define DEFINE_F(NAME, ASM)                                                                                   \
    __global__ void k_##NAME(unsigned long long* cyc, double* sink, int iters, int dep)                         \
    {                                                                                                         \
        float a[8];                                                                                           \
        for (int k = 0; k < 8; ++k) a[k] = 1.0f + 1e-3f * (threadIdx.x + k);                                    \
        const float c = 1.0000001f, d = 1e-9f;                                                                \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                           \
        if (dep) {                                                                                            \
            for (int i = 0; i < iters; ++i) {                                                                 \
                _Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile(ASM : "+v"(a[0]) : "v"(c), "v"(d)); \
            }                                                                                                 \
        } else {                                                                                              \
            for (int i = 0; i < iters; ++i) {                                                                 \
                _Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile(ASM : "+v"(a[k]) : "v"(c), "v"(d)); \
            }                                                                                                 \
        }                                                                                                     \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                           \
        float s = 0;                                                                                          \
        for (int k = 0; k < 8; ++k) s += a[k];                                                                \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                      \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;             \
    }

DEFINE_D(fma_f64, "v_fma_f64 %0, %0, %1, %2")
DEFINE_D(mul_f64, "v_mul_f64 %0, %0, %1")
DEFINE_D(add_f64, "v_add_f64 %0, %0, %2")
DEFINE_D(rcp_f64, "v_rcp_f64 %0, %0")
DEFINE_D(rsq_f64, "v_rsq_f64 %0, %0")
DEFINE_D(sqrt_f64, "v_sqrt_f64 %0, %0")
DEFINE_D(rndne_f64, "v_rndne_f64 %0, %0")
DEFINE_D(ldexp_f64, "v_ldexp_f64 %0, %0, 1")
DEFINE_D(max_f64, "v_max_f64 %0, %0, %1")
DEFINE_D(mov_b64, "v_mov_b64 %0, %1")
DEFINE_D(cmp_f64, "v_cmp_lt_f64 vcc, %0, %1")
DEFINE_F(fma_f32, "v_fma_f32 %0, %0, %1, %2")
DEFINE_F(mul_f32, "v_mul_f32 %0, %0, %1")
DEFINE_F(rcp_f32, "v_rcp_f32 %0, %0")
DEFINE_D(pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
DEFINE_D(pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
DEFINE_D(pk_add_f32, "v_pk_add_f32 %0, %0, %2")
DEFINE_F(cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
DEFINE_F(mov_b32, "v_mov_b32 %0, %1")
DEFINE_F(add_u32, "v_add_u32 %0, %0, %1")
DEFINE_F(exp_f32, "v_exp_f32 %0, %0")
DEFINE_F(cmp_i32, "v_cmp_gt_i32 vcc, %0, %1")
// a round trip double -> float -> double: two conversions per step
__global__ void k_cvt_pair(unsigned long long* cyc, double* sink, int iters, int dep)
{
    double a[8];
    float f[8];
    for (int k = 0; k < 8; ++k) a[k] = 1.0 + 1e-3 * (threadIdx.x + k);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = dep ? 0 : k;
            asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[j]) : "v"(a[j]));
            asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[j]) : "v"(f[j]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int k = 0; k < 8; ++k) s += a[k];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = (t1 - t0) / 2;
}

// a ds_read_b64 broadcast (every lane the same address) per step, the way the line kernel fetches per-line constants
__global__ void k_ds_read_b64(unsigned long long* cyc, double* sink, int iters, int dep)
{
    __shared__ double s[512];
    for (int k = threadIdx.x; k < 512; k += blockDim.x) s[k] = k;
    __syncthreads();
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = 0;
    int base = dep ? (threadIdx.x & 63) * 8 : 0;  // dep: per-lane addresses, else broadcast
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double v;
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(base), "i"(k * 8));
            asm volatile("s_waitcnt lgkmcnt(7)");
            a[k] += v;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double t = 0;
    for (int k = 0; k < 8; ++k) t += a[k];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = t;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

typedef void (*kern_t)(unsigned long long*, double*, int, int);

int main()
{
    const int n_cu = 256, iters = 2000;
    unsigned long long* cyc;
    double* sink;
    hipMalloc(&cyc, sizeof(unsigned long long) * n_cu * 8 * 4 * 16);
    hipMalloc(&sink, sizeof(double) * n_cu * 8 * 256 * 16);
    struct Item { const char* name; kern_t k; };
    const Item items[] = {
        {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_rcp_f64", k_rcp_f64}, {"v_rsq_f64", k_rsq_f64},
        {"v_sqrt_f64", k_sqrt_f64}, {"v_rndne_f64", k_rndne_f64}, {"v_ldexp_f64", k_ldexp_f64}, {"v_max_f64", k_max_f64},
        {"cvt f64<->f32", k_cvt_pair}, {"v_mov_b64", k_mov_b64}, {"v_cmp_lt_f64", k_cmp_f64},
        {"v_cndmask_b32", k_cndmask_b32}, {"v_fma_f32", k_fma_f32}, {"v_mul_f32", k_mul_f32}, {"v_rcp_f32", k_rcp_f32},
        {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_add_f32", k_pk_add_f32}, {"ds_read_b64", k_ds_read_b64},
        {"v_mov_b32", k_mov_b32}, {"v_add_u32", k_add_u32}, {"v_exp_f32", k_exp_f32},
        {"v_cmp_gt_i32", k_cmp_i32}, 
        
    };
    printf("%-16s %5s | cycles per wave-instruction (median over waves): independent x8 / dependent\n", "instruction", "");
    printf("%-16s %5s | %14s %14s %14s %14s\n", "", "", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD", "8 waves/SIMD");
    for (const Item& it : items) {
        printf("%-16s       |", it.name);
        for (int wps : {1, 2, 4, 8}) {
            double res[2];
            for (int dep = 0; dep < 2; ++dep) {
                // blocks of 256 threads = 1 wave per SIMD; wps blocks per CU
                const int blocks = n_cu * wps;
                hipLaunchKernelGGL(it.k, dim3(blocks), dim3(256), 0, 0, cyc, sink, iters, dep);
                hipDeviceSynchronize();
                hipLaunchKernelGGL(it.k, dim3(blocks), dim3(256), 0, 0, cyc, sink, iters, dep);
                hipDeviceSynchronize();
                std::vector<unsigned long long> h(blocks * 4);
                hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                std::sort(h.begin(), h.end());
                res[dep] = (double)h[h.size() / 2] / (iters * 8.0);
            }
            printf("  %5.1f / %5.1f", res[0], res[1]);
        }
        printf("\n");
    }
    return 0;
}
