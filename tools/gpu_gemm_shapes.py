"""The launch shapes of the condensed schedule's elimination and of the final inverse of the factor at cfg3's block size (n = 19 968,
141 eliminated + 15 kept tiles, leading blocks of a fifth): isolated TFLOP/s per launch (diagnostic; run through gpurun)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynadjust_amd.device import DeviceContext
ctx = DeviceContext(0)
f = ctx.lib.dnagpu_bench_gemm
f.restype = C.c_int
f.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.POINTER(C.c_double)] * 2
names = {0: "NT", 1: "NN", 2: "TN"}
km = {0: "full", 1: "k<=j", 2: "k>=j", 3: "k<=i", 4: "k>=i"}
cases = []
si, tj = 141, 15
while si > 0:
    h = si if si <= 12 else max(1, min(si, int(si * 0.2 + 0.5)))
    r = si - h + tj
    if r > 0 and h >= 8:
        cases.append(("panel  W = A X^T   ", 0, r, h, h * 128, 1, 0))
        cases.append(("update A -= W W^T  ", 0, r, r, h * 128, 0, 1))
        cases.append(("finish T = L X_bb  ", 1, r, h, h * 128, 2, 0))
        cases.append(("finish X = -X T    ", 1, r, h, r * 128, 3, 0))
    si -= h
tot_f = tot_t = 0.0
for (what, v, mt, nt, K, kmode, lower) in cases:
    ms = C.c_double(); fl = C.c_double()
    rc = f(ctx.h, v, mt, nt, K, kmode, lower, 3, C.byref(ms), C.byref(fl))
    assert rc == 0, ctx.lib.dnagpu_last_error(ctx.h)
    tiles = mt * (mt + 1) // 2 if lower else mt * nt
    tot_f += fl.value; tot_t += ms.value
    print(f"{what} {names[v]} mt={mt:4d} nt={nt:4d} K={K:6d} {km[kmode]:5s} lower={lower} tiles={tiles:6d} ({tiles/512:6.2f} waves): {ms.value:8.3f} ms  {fl.value/ms.value/1e9:6.2f} TFLOP/s", flush=True)
print(f"all: {tot_t:.2f} ms, {tot_f/tot_t/1e9:.2f} TFLOP/s")
