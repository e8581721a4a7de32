"""Per-variant throughput of the fp64 tile GEMM (diagnostic; run through gpurun)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynadjust_amd.device import DeviceContext
ctx = DeviceContext(0)
f = ctx.lib.dnagpu_bench_gemm
f.restype = C.c_int
f.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.POINTER(C.c_double)] * 2
names = {0: "NT", 1: "NN", 2: "TN"}
km = {0: "full", 1: "k<=j", 2: "k>=j", 3: "k<=i", 4: "k>=i"}
cases = [
    (0, 64, 64, 8192, 0, 0), (1, 64, 64, 8192, 0, 0), (2, 64, 64, 8192, 0, 0),
    (0, 118, 118, 15104, 0, 1),          # top-level SYRK at n=30000
    (0, 118, 117, 14976, 1, 0),          # W21 = A21 X11^T
    (1, 118, 117, 14976, 2, 0),          # T21 = W21 X11
    (1, 118, 117, 15104, 3, 0),          # X21 = -X22 T21
    (2, 128, 128, 16384, 4, 1),          # lauum
    (0, 16, 16, 2048, 0, 1), (0, 8, 8, 1024, 0, 1), (0, 4, 4, 512, 0, 1), (0, 1, 1, 128, 0, 0),
]
for (v, mt, nt, K, kmode, lower) in cases:
    ms = C.c_double(); fl = C.c_double()
    rc = f(ctx.h, v, mt, nt, K, kmode, lower, 3, C.byref(ms), C.byref(fl))
    assert rc == 0, ctx.lib.dnagpu_last_error(ctx.h)
    print(f"{names[v]} mt={mt:4d} nt={nt:4d} K={K:6d} {km[kmode]:5s} lower={lower}: {ms.value:9.3f} ms  {fl.value/ms.value/1e9:7.2f} TFLOP/s", flush=True)
