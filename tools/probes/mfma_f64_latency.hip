// Probe: latency of DEPENDENT v_mfma_f64_16x16x4_f64 (one accumulator chain) against 2 / 4 independent chains, one workgroup, 1 or 2 waves
// per SIMD -- what a 16 x 16 tile operation of the leaf kernel (4 dependent MFMAs) costs.  Shader clocks per MFMA (s_memtime).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f64_latency.hip -o build/mfma_f64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ __launch_bounds__(512) void chain(double* out, unsigned long long* clk, int iters, double a0, double b0) {
    d4 acc[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < CH; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[threadIdx.x >> 6] = t1 - t0;
}

template <int CH>
static void run(int threads) {
    double* out;
    unsigned long long* clk;
    hipMalloc(&out, 512 * 8);
    hipMalloc(&clk, 8 * 8);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(chain<CH>, dim3(1), dim3(threads), 0, 0, out, clk, iters, 1.0, 2.0);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    printf("%d chain(s), %d wave(s) per SIMD: %.1f clocks per MFMA per wave (wave 0)\n", CH, threads / 256, (double)h[0] / (iters * 4.0 * CH));
}

int main() {
    run<1>(256); run<2>(256); run<4>(256);
    run<1>(512); run<2>(512); run<4>(512);
    return 0;
}
