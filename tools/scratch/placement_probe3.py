"""Pair matrix of the two-grid passes X + Y over K work buffers (fixed sources): do the buffers fall into
classes such that a launch is fast iff its two work buffers share a class?  GPU box only."""
import ctypes as C
import importlib
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lib = importlib.import_module("21cmfast_amd").load(require_gpu=True)
lib.c21hip_split_floats.restype = C.c_size_t
lib.c21hip_split_floats.argtypes = [C.c_int] * 3
nf = lib.c21hip_split_floats(n, n, n)
lib.c21hip_split_filter_xy2.restype = C.c_int
lib.c21hip_split_filter_xy2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_float, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_float,
                                        C.c_int, C.c_int, C.c_int, C.c_void_p]
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
sa = torch.randn(nf, device="cuda", dtype=torch.float32)
sb = torch.randn(nf, device="cuda", dtype=torch.float32)
W = [torch.empty(nf, device="cuda", dtype=torch.float32) for _ in range(K)]


def t(i, j, reps=3):
    def run():
        assert lib.c21hip_split_filter_xy2(sa.data_ptr(), W[i].data_ptr(), 0, 0.0, sb.data_ptr(), W[j].data_ptr(), 0, 0.0,
                                           n, n, n, 1.5 * n, 1.5 * n, 10.0, 0, 0, 1, stream) == 0
    run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


M = [[t(i, j) if i != j else 0.0 for j in range(K)] for i in range(K)]
lo = min(M[i][j] for i in range(K) for j in range(K) if i != j)
print(f"n={n} K={K} fastest pair {lo:.3f} ms; matrix of t / fastest (rows: work a, columns: work b)")
for i in range(K):
    print("  " + " ".join("  -  " if i == j else f"{M[i][j] / lo:5.2f}" for j in range(K)))
ptrs = [w.data_ptr() for w in W]
print("  pointers: " + " ".join(f"{p:#x}" for p in ptrs))
# which single address bit b predicts "slow" as (ptr_i >> b & 1) == (ptr_j >> b & 1)?
pairs = [(i, j) for i in range(K) for j in range(K) if i != j]
slow = {(i, j): M[i][j] > 1.06 * lo for i, j in pairs}
for b in range(21, 47):
    same = sum(1 for (i, j) in pairs if ((ptrs[i] >> b) & 1) == ((ptrs[j] >> b) & 1) and slow[(i, j)])
    diff = sum(1 for (i, j) in pairs if ((ptrs[i] >> b) & 1) != ((ptrs[j] >> b) & 1) and not slow[(i, j)])
    agree = same + diff
    if agree >= 0.9 * len(pairs) or agree <= 0.1 * len(pairs):
        print(f"  bit {b}: 'slow iff equal' agrees on {agree} of {len(pairs)} pairs")
