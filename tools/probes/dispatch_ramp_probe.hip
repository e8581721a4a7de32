// How long does the GPU take to get one wave of tile-GEMM-shaped workgroups (256 threads, 72 KiB LDS, ~250 VGPRs, 2 per CU) running?
// Every workgroup records the time it starts and ends; the spread of the 512 start times is the ramp a launch pays before all CUs compute.
//   hipcc --offload-arch=gfx950 -O2 -o variants/dispatch_ramp_probe tools/probes/dispatch_ramp_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int LDS_DOUBLES, int ACC>
__global__ __launch_bounds__(256, 2) void shaped_kernel(long long* t_start, long long* t_end, long long hold_ticks, double* sink) {
    __shared__ double lds[LDS_DOUBLES];
    const long long t0 = wall_clock64();
    double acc[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    lds[threadIdx.x] = acc[1];
    __syncthreads();
    while (wall_clock64() - t0 < hold_ticks) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) acc[i] = acc[i] * 1.0000001 + lds[(threadIdx.x + i) & 255] * 1e-9;
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += acc[i];
    if (s == 1.2345) sink[0] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        t_start[blockIdx.x] = t0;
        t_end[blockIdx.x] = wall_clock64();
    }
}

int main() {
    int khz = 100000;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    const double us = 1000.0 / khz;
    long long *d0, *d1;
    double* sink;
    const int maxg = 4096;
    hipMalloc(&d0, maxg * 8);
    hipMalloc(&d1, maxg * 8);
    hipMalloc(&sink, 64);
    std::vector<long long> h0(maxg), h1(maxg);
    for (int variant = 0; variant < 4; ++variant)
    for (int grid : {512, 3584}) {
        for (double hold_us : {50.0, 500.0}) {
            for (int rep = 0; rep < 3; ++rep) {
                const long long ticks = (long long)(hold_us / us);
                if (variant == 0) hipLaunchKernelGGL((shaped_kernel<9216, 96>), dim3(grid), dim3(256), 0, 0, d0, d1, ticks, sink);
                if (variant == 1) hipLaunchKernelGGL((shaped_kernel<1024, 96>), dim3(grid), dim3(256), 0, 0, d0, d1, ticks, sink);
                if (variant == 2) hipLaunchKernelGGL((shaped_kernel<9216, 8>), dim3(grid), dim3(256), 0, 0, d0, d1, ticks, sink);
                if (variant == 3) hipLaunchKernelGGL((shaped_kernel<1024, 8>), dim3(grid), dim3(256), 0, 0, d0, d1, ticks, sink);
                hipDeviceSynchronize();
            }
            printf("%s  ", variant == 0 ? "72 KiB LDS, 96 acc" : variant == 1 ? " 8 KiB LDS, 96 acc" : variant == 2 ? "72 KiB LDS,  8 acc" : " 8 KiB LDS,  8 acc");
            hipMemcpy(h0.data(), d0, grid * 8, hipMemcpyDeviceToHost);
            hipMemcpy(h1.data(), d1, grid * 8, hipMemcpyDeviceToHost);
            std::vector<long long> s(h0.begin(), h0.begin() + grid);
            std::sort(s.begin(), s.end());
            const long long first = s[0];
            const long long last_end = *std::max_element(h1.begin(), h1.begin() + grid);
            const int w = std::min(grid, 512);
            printf("grid %4d, workgroups of %3.0f us: start of the 64th / 256th / %dth workgroup %6.1f / %6.1f / %6.1f us after the first; whole launch %7.1f us (ideal %7.1f)\n",
                   grid, hold_us, w, (s[63] - first) * us, (s[255] - first) * us, (s[w - 1] - first) * us, (last_end - first) * us,
                   hold_us * ((grid + 511) / 512));
        }
    }
    return 0;
}
