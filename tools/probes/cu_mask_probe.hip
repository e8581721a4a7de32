// Can a handful of CUs be kept free for the latency-bound kernels of the recursion (leaves, products of a few tiles) while the tile
// GEMMs of other chains fill the chip?  hipExtStreamCreateWithCUMask: where do the bits land on the 8 XCDs, are two masks disjoint in
// hardware, and how long does a one-workgroup kernel wait behind a chip full of long workgroups -- with and without the masks.
//   hipcc --offload-arch=gfx950 -O2 -o variants/cu_mask_probe tools/probes/cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <map>
#include <set>
#include <thread>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void where_kernel(uint32_t* out) {
    if (threadIdx.x == 0) {
        uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
        uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
        out[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xffff);
    }
}

// a workgroup shaped like one of the tile GEMM (256 threads, 72 KiB LDS, 2 per CU) that stays for `us` microseconds
__global__ __launch_bounds__(256, 2) void hold_kernel(long long ticks, double* sink) {
    __shared__ double lds[9216];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    long long t0 = wall_clock64();
    double a = lds[threadIdx.x];
    while (wall_clock64() - t0 < ticks) a = a * 1.0000001 + 1e-9;
    if (a == 12345.678) sink[0] = a;
}

// a workgroup shaped like the leaf: 512 threads, 78.5 KiB LDS
__global__ __launch_bounds__(512) void small_kernel(double* sink) {
    __shared__ double lds[10048];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (lds[(threadIdx.x + 1) & 511] == -1.0) sink[1] = 1.0;
}

static void describe(const char* name, const std::vector<uint32_t>& w) {
    std::map<int, std::set<int>> per_xcc;
    for (uint32_t v : w) per_xcc[(v >> 16) & 0xf].insert(v & 0xff00);    // cu_id [11:8], sh [12], se [15:13]
    printf("%s:", name);
    int total = 0;
    for (auto& kv : per_xcc) {
        printf("  xcc%d:%zu", kv.first, kv.second.size());
        total += (int)kv.second.size();
    }
    printf("   -> %d distinct CUs\n", total);
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("device %s, %d CUs, wall clock %d kHz\n", p.name, ncu, p.clockRate);
    int wc_khz = 100000;
    hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
    printf("wall_clock64 rate %d kHz\n", wc_khz);
    const int words = (ncu + 31) / 32;
    uint32_t* d_out;
    double* d_sink;
    CHECK(hipMalloc(&d_out, 8192 * 4));
    CHECK(hipMalloc(&d_sink, 64));
    std::vector<uint32_t> h(8192);

    for (int reserve : {8, 16}) {
        for (int layout = 0; layout < 2; ++layout) {
            // layout 0: the first `reserve` bits; layout 1: every (ncu / reserve)-th bit
            std::vector<uint32_t> small(words, 0), big(words, 0);
            for (int c = 0; c < ncu; ++c) {
                bool r = layout == 0 ? c < reserve : (c % (ncu / reserve) == 0);
                (r ? small : big)[c / 32] |= 1u << (c % 32);
            }
            hipStream_t ss, sb;
            CHECK(hipExtStreamCreateWithCUMask(&ss, words, small.data()));
            CHECK(hipExtStreamCreateWithCUMask(&sb, words, big.data()));
            printf("== reserve %d CUs, layout %s\n", reserve, layout ? "strided" : "first bits");
            hipLaunchKernelGGL(where_kernel, dim3(8192), dim3(64), 0, ss, d_out);
            CHECK(hipStreamSynchronize(ss));
            CHECK(hipMemcpy(h.data(), d_out, 8192 * 4, hipMemcpyDeviceToHost));
            describe("  small mask", h);
            std::set<uint32_t> s_small;
            for (uint32_t v : h) s_small.insert(v & 0xfff00 | (v & 0xf0000));
            hipLaunchKernelGGL(where_kernel, dim3(8192), dim3(64), 0, sb, d_out);
            CHECK(hipStreamSynchronize(sb));
            CHECK(hipMemcpy(h.data(), d_out, 8192 * 4, hipMemcpyDeviceToHost));
            describe("  big mask  ", h);
            int overlap = 0;
            std::set<uint32_t> s_big;
            for (uint32_t v : h) s_big.insert(v & 0xfff00 | (v & 0xf0000));
            for (uint32_t v : s_big) overlap += s_small.count(v);
            printf("  CUs in both: %d\n", overlap);
            // latency of a leaf-shaped kernel behind 2 x ncu resident long workgroups
            const long long ticks = (long long)wc_khz * 3;      // 3 ms
            for (int masked = 0; masked < 2; ++masked) {
                hipStream_t big_s = masked ? sb : nullptr, small_s = masked ? ss : nullptr;
                hipStream_t plain_b, plain_s;
                CHECK(hipStreamCreateWithFlags(&plain_b, hipStreamNonBlocking));
                CHECK(hipStreamCreateWithFlags(&plain_s, hipStreamNonBlocking));
                if (!masked) { big_s = plain_b; small_s = plain_s; }
                hipLaunchKernelGGL(small_kernel, dim3(1), dim3(512), 0, small_s, d_sink);     // warm
                CHECK(hipStreamSynchronize(small_s));
                hipLaunchKernelGGL(hold_kernel, dim3(2 * ncu), dim3(256), 0, big_s, ticks, d_sink);
                std::this_thread::sleep_for(std::chrono::microseconds(500));
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(small_kernel, dim3(1), dim3(512), 0, small_s, d_sink);
                CHECK(hipStreamSynchronize(small_s));
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                auto t1 = std::chrono::steady_clock::now();
                CHECK(hipStreamSynchronize(big_s));
                double hold_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                (void)t1;
                printf("  %s: 10 dependent one-workgroup kernels behind a full chip: %.0f us (the holders finished after %.0f us)\n",
                       masked ? "masked streams  " : "unmasked streams", us, hold_us);
                hipStreamDestroy(plain_b);
                hipStreamDestroy(plain_s);
            }
            hipStreamDestroy(ss);
            hipStreamDestroy(sb);
        }
    }
    // do two streams with the SAME mask run side by side?  64 workgroups of 2 ms on each of n streams: n x 2 ms = one queue behind them
    {
        std::vector<uint32_t> big(words, 0);
        for (int c = 8; c < ncu; ++c) big[c / 32] |= 1u << (c % 32);
        const long long ticks = (long long)wc_khz * 2;
        for (int variant = 0; variant < 3; ++variant) {
            hipStream_t st[4];
            for (int i = 0; i < 4; ++i) {
                if (variant == 0) {
                    CHECK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
                } else if (variant == 1) {
                    CHECK(hipExtStreamCreateWithCUMask(&st[i], words, big.data()));
                } else {
                    std::vector<uint32_t> m(big);          // same CUs, different vectors: bits beyond the device's CUs are ignored
                    m.push_back(1u << i);
                    CHECK(hipExtStreamCreateWithCUMask(&st[i], (uint32_t)m.size(), m.data()));
                }
            }
            for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(hold_kernel, dim3(8), dim3(256), 0, st[i], 1000LL, d_sink);
            for (int i = 0; i < 4; ++i) CHECK(hipStreamSynchronize(st[i]));
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(hold_kernel, dim3(64), dim3(256), 0, st[i], ticks, d_sink);
            for (int i = 0; i < 4; ++i) CHECK(hipStreamSynchronize(st[i]));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("4 streams, %s: 4 x (64 workgroups of 2 ms) took %.0f us\n",
                   variant == 0 ? "unmasked" : variant == 1 ? "one mask for all" : "same CUs, distinct mask vectors", us);
            for (int i = 0; i < 4; ++i) hipStreamDestroy(st[i]);
        }
    }
    return 0;
}
