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
    printf("shader clocks (100 MHz constant clock on gfx9: x10 ns)\n");
    printf("load            %8.0f\n", d(0, 1));
    double diag = 0, panel = 0, trail = 0;
    for (int kb = 0; kb < 8; ++kb) {
        int prev = kb == 0 ? 1 : 4 + 3 * (kb - 1);
        printf("A kb=%d  diag %6.0f  panel %6.0f  trailing %6.0f\n", kb, d(prev, 2 + 3 * kb), d(2 + 3 * kb, 3 + 3 * kb), d(3 + 3 * kb, 4 + 3 * kb));
        diag += d(prev, 2 + 3 * kb);
        panel += d(2 + 3 * kb, 3 + 3 * kb);
        trail += d(3 + 3 * kb, 4 + 3 * kb);
    }
    printf("phase A: diag %6.0f  panel %6.0f  trailing %6.0f\n", diag, panel, trail);
    for (int kb = 0; kb < 8; ++kb) printf("B kb=%d  %6.0f\n", kb, d(kb == 0 ? 25 : 25 + kb, 26 + kb));
    printf("phase B total   %8.0f\n", d(25, 33));
    printf("store           %8.0f\n", d(33, 34));
    printf("whole kernel    %8.0f\n", d(0, 34));
    return 0;
}
