"""Does the leading dimension (a multiple of 128 doubles = 1 KiB everywhere in the product) cost L2 set conflicts?  The tile GEMM on the
recursion's shapes at n = 19 968 with the leading dimension padded by 0 / 16 / 32 / 80 doubles (diagnostic; run through gpurun; under
rocprofv3 --pmc FETCH_SIZE the same runs give the fabric reads per variant)."""
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
    (2, 156, 156, 19968, 4, 1),          # lauum at cfg3's block size
    (0, 78, 78, 9984, 0, 1),             # top-level SYRK
    (0, 78, 78, 9984, 1, 0),             # W21 = A21 X11^T
    (1, 78, 78, 9984, 2, 0),             # T21 = W21 X11
    (1, 78, 78, 9984, 3, 0),             # X21 = -X22 T21
]
pads = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "16", "32", "80"])]
for pad in pads:
    os.environ["DNAGPU_BENCH_LDPAD"] = str(pad)
    for (v, mt, nt, K, kmode, lower) in cases:
        ms = C.c_double(); fl = C.c_double()
        rc = f(ctx.h, v, mt, nt, K, kmode, lower, 3, C.byref(ms), C.byref(fl))
        assert rc == 0, ctx.lib.dnagpu_last_error(ctx.h)
        print(f"pad {pad:3d} {names[v]} mt={mt:4d} nt={nt:4d} K={K:6d} {km[kmode]:5s} lower={lower}: {ms.value:9.3f} ms  {fl.value/ms.value/1e9:7.2f} TFLOP/s", flush=True)
