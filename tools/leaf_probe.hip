// Where the leaf kernel's time goes: compiles dynadjust_amd/csrc/leaf_kernel.hip with its probe points on and prints the shader-clock
// distance between them (thread 0's view), for a well-conditioned 128x128 tile.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDNAGPU_LEAF_PROBE -Idynadjust_amd/csrc -Iinclude tools/leaf_probe.hip -o gpurun_out/leaf_probe
#include "../dynadjust_amd/csrc/leaf_kernel.hip"
#include <cstdio>
#include <vector>

int main() {
    const int n = 128;
    std::vector<double> A((size_t)n * n);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) A[(size_t)j * n + i] = (i == j) ? n + 1.0 : 1.0 / (1.0 + (i > j ? i - j : j - i));
    double *dA, *dX;
    int* dinfo;
    hipMalloc(&dA, A.size() * 8);
    hipMalloc(&dX, A.size() * 8);
    hipMalloc(&dinfo, 4);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    int big = 1 << 30;
    hipMemcpy(dinfo, &big, 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        for (int k = 0; k < 100; ++k) dnagpu::launch_leaf(dA, n, dX, n, 0, dinfo, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("100 leaves back to back: %.1f us each\n", ms * 10.0);
    }
    unsigned long long p[64];
    hipMemcpyFromSymbol(p, HIP_SYMBOL(dnagpu::leaf_probe), sizeof(p));
    auto d = [&](int a, int b) { return (double)(p[b] - p[a]); };
    printf("shader clocks (s_memtime; 67 200 of them are the 29 us of a leaf: ~2.3 GHz)\n");
    // (the overlapped schedule of round 4: leaf_body.h, potrf_trtri_tile_overlapped)
    printf("block column 0 in      %8.0f\n", d(0, 1));
    printf("diag 0 | rest in       %8.0f\n", d(1, 2));
    // probe 3 + 2 kb is taken by wave 1 (after the seven other waves have met), the others by wave 0
    double total = 0;
    for (int kb = 0; kb < 8; ++kb) {
        printf("kb=%d  panel + M(:, %d), waves 1..7 met %6.0f   whole step (diag %d | trailing + phase B step %d + row block %d out) %6.0f\n", kb, kb - 1,
               d(2 + 2 * kb, 3 + 2 * kb), kb + 1, kb, kb - 1, d(2 + 2 * kb, 4 + 2 * kb));
        total += d(2 + 2 * kb, 4 + 2 * kb);
    }
    printf("steps %8.0f\n", total);
    printf("row block 7 out        %8.0f\n", d(18, 20));
    printf("whole kernel           %8.0f\n", d(0, 20));
    return 0;
}
