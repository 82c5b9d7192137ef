"""How many classes are there?  K work spectra spread over most of the HBM (spacers in between), matrix of the
two-grid pass Y in place on (i, j).  GPU box only."""
import ctypes as C
import importlib
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
gap = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lib = importlib.import_module("21cmfast_amd").load(require_gpu=True)
lib.c21hip_split_floats.restype = C.c_size_t
lib.c21hip_split_floats.argtypes = [C.c_int] * 3
nf = lib.c21hip_split_floats(n, n, n)
lib.c21hip_probe_pass_y2.restype = C.c_int
lib.c21hip_probe_pass_y2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
W, spacers = [], []
for i in range(K):
    W.append(torch.zeros(nf, device="cuda", dtype=torch.float32))
    try:
        spacers.append(torch.empty(gap << 30, device="cuda", dtype=torch.uint8))
    except Exception:
        break


def y2(i, j):
    ms = C.c_float()
    assert lib.c21hip_probe_pass_y2(W[i].data_ptr(), W[j].data_ptr(), n, n, n, 4, C.byref(ms), stream) == 0
    return ms.value * 1e3


K = len(W)
M = [[y2(i, j) if i != j else 0. for j in range(K)] for i in range(K)]
lo = min(M[i][j] for i in range(K) for j in range(K) if i != j)
print(f"n={n}, {K} buffers {gap} GB apart: two-grid pass Y on (i, j), t / fastest ({lo:.1f} us)")
for i in range(K):
    print("  " + " ".join("  -  " if i == j else f"{M[i][j] / lo:5.2f}" for j in range(K)))
# classes: connected components of "slow together"
cls = list(range(K))
for i in range(K):
    for j in range(K):
        if i != j and M[i][j] > 1.06 * lo:
            a, b = cls[i], cls[j]
            cls = [a if c == b else c for c in cls]
print("classes:", cls, "->", len(set(cls)))
