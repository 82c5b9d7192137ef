"""Do out-of-place kernels care where source and destination sit?  K buffers allocated one after the other;
matrix of (a) a plain device copy i -> j, (b) single-grid passes X + Y with source i and work j.  GPU box only."""
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
lib.c21hip_split_filter_xy.restype = C.c_int
lib.c21hip_split_filter_xy.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                       C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p]
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
W = [torch.randn(nf, device="cuda", dtype=torch.float32) for _ in range(K)]


def timed(fn, reps=3):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def show(name, M):
    lo = min(M[i][j] for i in range(K) for j in range(K) if i != j)
    print(f"{name}: fastest {lo:.3f} ms; t / fastest (rows: source, columns: destination)")
    for i in range(K):
        print("  " + " ".join("  -  " if i == j else f"{M[i][j] / lo:5.2f}" for j in range(K)))


show("plain copy", [[timed(lambda: W[j].copy_(W[i])) if i != j else 0. for j in range(K)] for i in range(K)])
show("single-grid passes X + Y", [[timed(lambda: lib.c21hip_split_filter_xy(W[i].data_ptr(), W[j].data_ptr(), n, n, n, 1.5 * n, 1.5 * n, 0,
                                                                        10.0, 0.0, 0, stream)) if i != j else 0.
                                    for j in range(K)] for i in range(K)])
