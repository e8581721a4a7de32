// Probe: issue-rate ceiling of v_mfma_f64_16x16x4_f64 on gfx950 with the GEMM kernel's occupancy
// (2 workgroups x 4 waves per CU = 2 waves per SIMD, 16 independent accumulators per wave) and no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f64_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void mfma_loop(double* out, int iters, double a0, double b0) {
    d4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int W>
static void run(const char* label, int grid, size_t lds_pad) {
    double* out;
    hipMalloc(&out, (size_t)grid * 256 * sizeof(double));
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_loop<W>, dim3(grid), dim3(256), lds_pad, 0, out, iters, 1.0, 2.0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)grid * 4 * iters * 16 * 2048.0;
        printf("%s grid %d: %.3f ms  %.2f TFLOP/s\n", label, grid, ms, flops / ms / 1e9);
    }
    hipFree(out);
}

int main() {
    run<2>("2 waves/SIMD (2 wg/CU)", 512, 0);
    run<2>("2 waves/SIMD, 4 rounds", 2048, 0);
    run<1>("1 wave/SIMD  (1 wg/CU, 100 KB LDS pad)", 256, 100 * 1024);
    run<2>("4 waves/SIMD (4 wg/CU)", 1024, 0);
    return 0;
}
