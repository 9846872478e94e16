// Checks on the GPU that sdx::exp_neg(tau) returns the bits of exp(-tau) (ROCm device library) over the range the
// formal solution uses it on.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../stardis_amd/csrc exp_neg_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "sdx_math.h"
__global__ void k(uint64_t n, unsigned long long* bad, double* worst)
{
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    // log-uniform over [1e-6, 700) plus a fine linear sweep
    const double u = (double)i / (double)n;
    const double tau = (i & 1) ? exp(-13.8 + u * 20.35) : u * 60.0;
    const double a = sdx::exp_neg(tau), b = exp(-tau);
    if (a != b) {
        atomicAdd(bad, 1ull);
        worst[0] = tau;
    }
}
int main()
{
    unsigned long long* bad; double* worst;
    hipMalloc(&bad, 8); hipMalloc(&worst, 8); hipMemset(bad, 0, 8); hipMemset(worst, 0, 8);
    const uint64_t n = 1ull << 28;
    hipLaunchKernelGGL(k, dim3((unsigned)(n / 256)), dim3(256), 0, 0, n, bad, worst);
    unsigned long long h = 0; double w = 0;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&w, worst, 8, hipMemcpyDeviceToHost);
    printf("exp_neg vs exp(-tau): %llu of %llu differ (last differing tau %.17g)\n", h, (unsigned long long)n, w);
    return h != 0;
}
