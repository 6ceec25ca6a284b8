#!/usr/bin/env python3
"""Developer sweep of the stand-alone GEMV kernel: us and achieved TB/s (int8 + scale bytes) per shape/tile/grid."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
import qwen3_rs_amd as q3
L = q3.load_library()
L.q3_dev_bench_gemv.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.POINTER(C.c_float), C.POINTER(C.c_int32)]
def bench(n, d, G=64, wg=0, ru=0, ju=0, reps=50):
    us = C.c_float(0); used = (C.c_int32 * 3)()
    rc = L.q3_dev_bench_gemv(n, d, G, wg, ru, ju, reps, 0, C.byref(us), used)
    if rc != 0:
        return None
    by = n * d + 4 * (n * d // G)
    return us.value, by / us.value / 1e6, tuple(used)
shapes = [("lm_head 0.6B", 1024, 151936), ("8B wq", 4096, 4096), ("8B w1", 4096, 12288), ("8B w2", 12288, 4096),
          ("4B w1", 2560, 9728), ("4B w2", 9728, 2560), ("lm_head 8B", 4096, 151936)]
if len(sys.argv) > 1 and sys.argv[1] == "lm":
    shapes = shapes[:1]
for name, n, d in shapes:
    for wg in (1, 2, 3, 4, 6, 8):
        for ru, ju in ((0, 0),) if name != "lm_head 0.6B" else ((8, 1), (4, 1), (2, 1)):
            r = bench(n, d, 64, wg, ru, ju, reps=30 if d > 100000 else 100)
            if r: print(f"{name:14s} n={n:5d} d={d:6d} wg/cu={wg} tile={r[2][0]}x{r[2][1]} grid={r[2][2]:5d}  {r[0]:8.2f} us  {r[1]:6.2f} TB/s", flush=True)
