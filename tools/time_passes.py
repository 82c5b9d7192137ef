"""Per-pass timings through the c21hip_bench_pass hook (diagnostic; GPU box only)."""
import ctypes as C
import importlib
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lib = importlib.import_module("21cmfast_amd").load(require_gpu=True)
lib.c21hip_bench_pass.restype = C.c_int
lib.c21hip_bench_pass.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                  C.c_double, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
names = {0: "pass X 2 grids, windows (fa,fb)", 1: "pass Y 2 grids", 2: "fused pass Z",
         3: "plain pass Z (1 grid)", 4: "window tables", 5: "pass X 2 grids, no window",
         6: "pass X 2 grids, TWO radii", 7: "pass X 2 grids, W evaluated",
         8: "pass X TWO radii, W evaluated"}
for fa, fb in ((0, 3), (0, 0)):
    for kind in (0, 5, 7, 1, 2, 3, 4) + ((6, 8) if n < 1024 else ()):
        ms = C.c_float()
        st = lib.c21hip_bench_pass(kind, n, fa, fb, 12.0, 37.0, 1.5 * n, 20, stream, C.byref(ms))
        print(f"n={n} filters=({fa},{fb}) kind {kind} {names[kind]:34s} st={st} {ms.value*1e3:8.1f} us")
