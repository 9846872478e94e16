// How many workgroups of a given size and LDS footprint does an MI355X CU really hold at once?  Each block records its
// start time (constant 100 MHz clock) and spins ~20 us; blocks that start "late" had to wait for a slot.
// hipcc --offload-arch=gfx950 -O2 occupancy_probe.hip -o occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void spin(long long* start, int lds_doubles)
{
    extern __shared__ double s[];
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) start[blockIdx.x] = t0;
    if (lds_doubles > 0) s[threadIdx.x % lds_doubles] = (double)t0;
    while (wall_clock64() - t0 < 2000) {}  // 20 us
    if (lds_doubles > 0 && s[0] == 1.2345) start[blockIdx.x] = 0;
}
template <int NV>
__global__ __launch_bounds__(1024) void spin_regs(long long* start, int lds_doubles)
{
    extern __shared__ double s[];
    __shared__ double fixed[4408];  // 35264 B static, like the pre-pass
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) start[blockIdx.x] = t0;
    double v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = (double)(threadIdx.x * (k + 1)) + (double)t0;
    fixed[threadIdx.x] = v[0];
    __syncthreads();
    while (wall_clock64() - t0 < 2000) {
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = v[k] * 1.0000001 + fixed[(threadIdx.x + k) & 1023];
    }
    double acc = 0;
#pragma unroll
    for (int k = 0; k < NV; ++k) acc += v[k];
    if (acc == 1.2345) start[blockIdx.x] = 0;
    if (lds_doubles > 0 && s[0] == 1.2345) start[blockIdx.x] = 0;
}
template <int NV>
static void run_regs(long long* d, int blocks, int n_cu)
{
    hipLaunchKernelGGL(spin_regs<NV>, dim3(blocks), dim3(1024), 80, 0, d, 10);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), d, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    const long long t0 = *std::min_element(h.begin(), h.end());
    int first = 0;
    for (auto t : h) first += (t - t0) < 500;
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void*)spin_regs<NV>);
    printf("1024 threads, 35264 B static LDS + barrier, %d live doubles (numRegs %d): %d of %d at once = %.2f per CU\n", NV, fa.numRegs, first,
           blocks, (double)first / n_cu);
}
int main()
{
    int n_cu = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = n_cu * 4;
    long long* d;
    hipMalloc(&d, blocks * sizeof(long long));
    const int cases[][2] = {{1024, 0}, {1024, 35328}, {1024, 16384}, {512, 0}, {512, 17664}, {512, 35328}, {256, 50000}, {256, 0}, {64, 0}, {64, 5120}};
    for (auto& c : cases) {
        hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipLaunchKernelGGL(spin, dim3(blocks), dim3(c[0]), c[1], 0, d, c[1] / 8);
        hipDeviceSynchronize();
        std::vector<long long> h(blocks);
        hipMemcpy(h.data(), d, blocks * sizeof(long long), hipMemcpyDeviceToHost);
        const long long t0 = *std::min_element(h.begin(), h.end());
        int first = 0;
        for (auto t : h) first += (t - t0) < 500;  // started within 5 us of the first block
        printf("threads %4d  LDS %6d B: %4d of %d blocks start at once = %.2f per CU (%.1f waves per CU)\n", c[0], c[1], first, blocks,
               (double)first / n_cu, (double)first / n_cu * c[0] / 64);
    }
    run_regs<2>(d, blocks, n_cu);
    run_regs<8>(d, blocks, n_cu);
    run_regs<16>(d, blocks, n_cu);
    run_regs<20>(d, blocks, n_cu);
    run_regs<28>(d, blocks, n_cu);
    return 0;
}
