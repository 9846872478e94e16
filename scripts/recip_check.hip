// Accuracy of 1/d from v_rcp_f64 refined by (a) two Newton steps and (b) one cubically convergent step, against IEEE division.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off recip_check.hip -o recip_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
__device__ double recip_nr2(double d) { double r = __builtin_amdgcn_rcp(d); double e = fma(-d, r, 1.0); r = fma(r, e, r); e = fma(-d, r, 1.0); return fma(r, e, r); }
__device__ double recip_cubic(double d) { const double r = __builtin_amdgcn_rcp(d); const double e = fma(-d, r, 1.0); return fma(r, fma(e, e, e), r); }
__global__ void k(uint64_t n, double* worst, unsigned long long* diff)
{
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32;
    const double u = (double)(s >> 11) / 9007199254740992.0;
    const double d = exp2(-60.0 + 160.0 * ((double)(i % 1000) / 1000.0)) * (1.0 + u);  // 2^-60 .. 2^100
    const double ref = 1.0 / d;
    const double a = recip_nr2(d), b = recip_cubic(d);
    const double ea = fabs(a - ref) / ref, eb = fabs(b - ref) / ref;
    // atomic max on doubles >= 0 via integer compare
    atomicMax((unsigned long long*)&worst[0], __double_as_longlong(ea));
    atomicMax((unsigned long long*)&worst[1], __double_as_longlong(eb));
    if (a != ref) atomicAdd(&diff[0], 1ull);
    if (b != ref) atomicAdd(&diff[1], 1ull);
    if (a != b) atomicAdd(&diff[2], 1ull);
}
int main()
{
    double* w; unsigned long long* d;
    hipMalloc(&w, 16); hipMalloc(&d, 24); hipMemset(w, 0, 16); hipMemset(d, 0, 24);
    const uint64_t n = 1ull << 28;
    hipLaunchKernelGGL(k, dim3((unsigned)(n / 256)), dim3(256), 0, 0, n, w, d);
    double hw[2]; unsigned long long hd[3];
    hipMemcpy(hw, w, 16, hipMemcpyDeviceToHost); hipMemcpy(hd, d, 24, hipMemcpyDeviceToHost);
    printf("two Newton steps: max rel err %.3e, %llu of %llu differ from IEEE division\n", hw[0], hd[0], (unsigned long long)n);
    printf("one cubic step  : max rel err %.3e, %llu differ from IEEE division; %llu differ from the two-step value\n", hw[1], hd[1], hd[2]);
    return 0;
}
