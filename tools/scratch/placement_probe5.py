"""Single-grid pass Y from buffer i to buffer j (i == j: in place, what the product runs) over K buffers that are
25 GB apart in allocation order: is an out-of-place pass Y into another region of the HBM faster?  GPU box only."""
import ctypes as C
import importlib
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lib = importlib.import_module("21cmfast_amd").load(require_gpu=True)
lib.c21hip_split_floats.restype = C.c_size_t
lib.c21hip_split_floats.argtypes = [C.c_int] * 3
nf = lib.c21hip_split_floats(n, n, n)
lib.c21hip_probe_pass_y1.restype = C.c_int
lib.c21hip_probe_pass_y1.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
lib.c21hip_probe_pass_y2.restype = C.c_int
lib.c21hip_probe_pass_y2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
W, spacers = [], []
for i in range(K):
    W.append(torch.zeros(nf, device="cuda", dtype=torch.float32))
    spacers.append(torch.empty(18 << 30, device="cuda", dtype=torch.uint8))


def y1(i, j):
    ms = C.c_float()
    assert lib.c21hip_probe_pass_y1(W[i].data_ptr(), W[j].data_ptr(), n, n, n, 5, C.byref(ms), stream) == 0
    return ms.value


def y2(i, j):
    ms = C.c_float()
    assert lib.c21hip_probe_pass_y2(W[i].data_ptr(), W[j].data_ptr(), n, n, n, 5, C.byref(ms), stream) == 0
    return ms.value


print(f"n={n}: single-grid pass Y, source i (row) -> destination j (column), us; diagonal = in place")
for i in range(K):
    print("  " + " ".join(f"{y1(i, j) * 1e3:6.1f}" for j in range(K)))
print("two-grid pass Y in place on (i, j), us (the classes)")
for i in range(K):
    print("  " + " ".join("   -  " if i == j else f"{y2(i, j) * 1e3:6.1f}" for j in range(K)))

# two grids, out of place: classes from the matrix above (buffer 0's class = A)
lib.c21hip_probe_pass_y2o.restype = C.c_int
lib.c21hip_probe_pass_y2o.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.POINTER(C.c_float), C.c_void_p]


def y2o(sa, da, sb, db):
    ms = C.c_float()
    assert lib.c21hip_probe_pass_y2o(W[sa].data_ptr(), W[da].data_ptr(), W[sb].data_ptr(), W[db].data_ptr(), n, n, n, 5,
                                     C.byref(ms), stream) == 0
    return ms.value * 1e3


ref = [y2(0, j) if j else 0. for j in range(K)]
lo = min(t for t in ref[1:])
A = [0] + [j for j in range(1, K) if ref[j] > 1.06 * lo]
B = [j for j in range(1, K) if ref[j] <= 1.06 * lo]
print("class A", A, "class B", B)
if len(A) >= 2 and len(B) >= 2:
    print("two grids, us: in place on (A0, B0) %.1f | a: A0->B0, b: B1->A1 %.1f | a: A0->A1, b: B0->B1 %.1f | a: A0->B0, b: A1->B1 %.1f"
          % (y2o(A[0], A[0], B[0], B[0]), y2o(A[0], B[0], B[1], A[1]), y2o(A[0], A[1], B[0], B[1]), y2o(A[0], B[0], A[1], B[1])))
if len(A) >= 4:
    print("   all four in A: a: A0->A1, b: A2->A3 %.1f" % y2o(A[0], A[1], A[2], A[3]))
