// Probe: what costs the GEMM main loop its last 10 %?  The loop of gemm_f64_kernel (4 waves x 64x64, BK = 16 slab =
// 4 k-steps x 16 MFMAs) rebuilt step by step on synthetic data:
//   level 0: MFMAs only (register operands)                      level 1: + fragment ds_read_b64 per k-step
//   level 2: + one __syncthreads per slab                        level 3: + 8 ds_write_b128 per slab (register data)
//   level 4: + 8 global_load_dwordx4 per slab feeding the ds_writes (L2-resident source)
//   level 5: level 4 with sched_group_barrier spreading the loads between the MFMAs
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f64_feed.hip -o variants/mfma_feed && variants/mfma_feed
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

template <int LEVEL>
__global__ __launch_bounds__(256, 2) void loop(double* out, const double* src, int slabs) {
    __shared__ __attribute__((aligned(16))) double lds[4 * 16 * 144];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
    for (int i = tid; i < 4 * 16 * 144; i += 256) lds[i] = 1.0 + i * 1e-6;
    __syncthreads();
    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
    double af[4] = {1.0, 2.0, 3.0, 4.0}, bf[4] = {1.5, 2.5, 3.5, 4.5};
    d2 g[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) g[q] = (d2){1.0 * q, 2.0 * q};
    const double* gsrc = src + (size_t)(blockIdx.x % 64) * 4096 + tid * 2;
    for (int t = 0; t < slabs; ++t) {
        const double* As = lds + (t & 1) * 2 * 16 * 144;
        const double* Bs = As + 16 * 144;
        if (LEVEL >= 4) {
#pragma unroll
            for (int q = 0; q < 8; ++q) g[q] = *reinterpret_cast<const d2*>(gsrc + q * 512 + (t & 7) * 4096 * 64);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (LEVEL >= 1) {
                int k = kk * 4 + (lane >> 4);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) af[mi] = As[k * 144 + wm * 64 + mi * 16 + (lane & 15)];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) bf[ni] = Bs[k * 144 + wn * 64 + ni * 16 + (lane & 15)];
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[ni], af[mi], acc[mi][ni], 0, 0, 0);
        }
        if (LEVEL >= 3) {
            double* An = lds + ((t & 1) ^ 1) * 2 * 16 * 144;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                int idx = tid + 256 * q, k = (idx / 64) & 15, r2 = idx % 64;
                *reinterpret_cast<d2*>(An + (q >= 4 ? 16 * 144 : 0) + k * 144 + 2 * r2) = g[q];
            }
        }
        if (LEVEL >= 5) {
            // level 5 = level 4 with the loads spread over the slab: 1 global load per 8 MFMAs
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
        }
        if (LEVEL >= 2) __syncthreads();
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + tid] = s;
}

template <int LEVEL>
static void run(double* out, const double* src) {
    const int grid = 2048, slabs = 512;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(loop<LEVEL>, dim3(grid), dim3(256), 0, 0, out, src, slabs);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double flops = (double)grid * 4 * slabs * 64 * 2048.0;
    printf("level %d: %.3f ms  %.2f TFLOP/s\n", LEVEL, best, flops / best / 1e9);
}

int main() {
    double *out, *src;
    hipMalloc(&out, 2048 * 256 * sizeof(double));
    hipMalloc(&src, (size_t)64 * 4096 * 64 * 8 * sizeof(double) + (1 << 20));
    hipMemset(src, 0, (size_t)64 * 4096 * 64 * 8 * sizeof(double) + (1 << 20));
    run<0>(out, src);
    run<1>(out, src);
    run<2>(out, src);
    run<3>(out, src);
    run<4>(out, src);
    run<5>(out, src);
    return 0;
}
