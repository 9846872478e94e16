// Micro-benchmarks on gfx950: sustained fp64 FMA rate, v_rcp_f64 rate + accuracy, exp() rate.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/fp64_peak scripts/fp64_peak.hip && /tmp/fp64_peak
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

template <int ILP>
__global__ void k_fma(double* out, int iters, double a, double b)
{
    double x[ILP];
    for (int k = 0; k < ILP; ++k) x[k] = threadIdx.x * 1e-3 + k;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int k = 0; k < ILP; ++k) x[k] = fma(x[k], a, b);
    double s = 0;
    for (int k = 0; k < ILP; ++k) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_rcp(double* out, int iters, double a)
{
    double x = 1.0 + threadIdx.x * 1e-3, y = 2.0 + threadIdx.x * 1e-3, z = 3.0 + threadIdx.x * 1e-3, w = 4.0 + threadIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) {
        x = __builtin_amdgcn_rcp(x) + a; y = __builtin_amdgcn_rcp(y) + a; z = __builtin_amdgcn_rcp(z) + a; w = __builtin_amdgcn_rcp(w) + a;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y + z + w;
}

__global__ void k_exp(double* out, int iters, double a)
{
    double x = -1.0 - threadIdx.x * 1e-3, y = -2.0 - threadIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) { x = exp(x) - a; y = exp(y) - a; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}

__global__ void k_div(double* out, int iters, double a)
{
    double x = 1.5 + threadIdx.x * 1e-3, y = 2.5 + threadIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) { x = a / x + 1.0; y = a / y + 1.0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}

__global__ void k_rcp_acc(const double* in, double* raw, double* nr1, double* nr2, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = in[i];
    double r = __builtin_amdgcn_rcp(d);
    raw[i] = r;
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    nr1[i] = r;
    e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    nr2[i] = r;
}

template <typename F>
double time_ms(F f, int reps = 5)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    double* out;
    hipMalloc(&out, sizeof(double) * 256 * 4096 * 4);
    const int iters = 4096;
    for (int blocks_per_cu : {1, 2, 4, 8}) {
        const int blocks = 256 * blocks_per_cu;
        double ms1 = time_ms([&] { k_fma<1><<<blocks, 256>>>(out, iters, 1.0000001, 1e-9); });
        double ms4 = time_ms([&] { k_fma<4><<<blocks, 256>>>(out, iters / 4, 1.0000001, 1e-9); });
        double fl = 2.0 * blocks * 256.0 * iters;
        printf("fma f64: %d waves/SIMD  ILP1 %.2f TFLOP/s   ILP4 %.2f TFLOP/s\n", blocks_per_cu, fl / ms1 * 1e-9, fl / ms4 * 1e-9);
    }
    {
        const int blocks = 256 * 8;
        double ms = time_ms([&] { k_rcp<<<blocks, 256>>>(out, iters, 1.0); });
        printf("v_rcp_f64 (+1 add): %.1f G rcp/s  (=> %.1f cycles per wave-instr pair per SIMD at 2.4 GHz)\n", 4.0 * blocks * 256.0 * iters / ms * 1e-6,
               1024 * 2.4e9 / (4.0 * blocks * 256.0 * iters / (ms * 1e-3) / 64));
        ms = time_ms([&] { k_exp<<<blocks, 256>>>(out, iters / 4, 0.5); });
        printf("exp(double): %.1f G exp/s (%.1f cycles per wave-call per SIMD)\n", 2.0 * blocks * 256.0 * (iters / 4) / ms * 1e-6,
               1024 * 2.4e9 / (2.0 * blocks * 256.0 * (iters / 4) / (ms * 1e-3) / 64));
        ms = time_ms([&] { k_div<<<blocks, 256>>>(out, iters / 4, 3.0); });
        printf("IEEE div (+1 add): %.1f G div/s (%.1f cycles per wave-call per SIMD)\n", 2.0 * blocks * 256.0 * (iters / 4) / ms * 1e-6,
               1024 * 2.4e9 / (2.0 * blocks * 256.0 * (iters / 4) / (ms * 1e-3) / 64));
    }
    {
        const int n = 1 << 20;
        std::vector<double> h(n), raw(n), n1(n), n2(n);
        for (int i = 0; i < n; ++i) h[i] = std::exp((i / (double)n) * 100.0 - 50.0) * (1.0 + (i % 977) * 1e-3);
        double *d, *r0, *r1, *r2;
        hipMalloc(&d, n * 8); hipMalloc(&r0, n * 8); hipMalloc(&r1, n * 8); hipMalloc(&r2, n * 8);
        hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
        k_rcp_acc<<<n / 256, 256>>>(d, r0, r1, r2, n);
        hipMemcpy(raw.data(), r0, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(n1.data(), r1, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(n2.data(), r2, n * 8, hipMemcpyDeviceToHost);
        double e0 = 0, e1 = 0, e2 = 0;
        for (int i = 0; i < n; ++i) {
            long double t = 1.0L / (long double)h[i];
            e0 = std::fmax(e0, (double)fabsl(((long double)raw[i] - t) / t));
            e1 = std::fmax(e1, (double)fabsl(((long double)n1[i] - t) / t));
            e2 = std::fmax(e2, (double)fabsl(((long double)n2[i] - t) / t));
        }
        printf("v_rcp_f64 max rel err: raw %.3e, +1 Newton %.3e, +2 Newton %.3e\n", e0, e1, e2);
    }
    return 0;
}
