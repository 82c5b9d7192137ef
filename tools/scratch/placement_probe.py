"""Do the two speeds of the line passes belong to the buffers (placement) or to the process?  One process,
several (source, work) spectrum pairs allocated one after the other and all kept; passes X + Y (no window)
timed on each pair, then on the first pairs again.  GPU box only."""
import ctypes as C
import importlib
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 6
lib = importlib.import_module("21cmfast_amd").load(require_gpu=True)
lib.c21hip_split_floats.restype = C.c_size_t
lib.c21hip_split_floats.argtypes = [C.c_int] * 3
nf = lib.c21hip_split_floats(n, n, n)
lib.c21hip_split_filter_xy.restype = C.c_int
lib.c21hip_split_filter_xy.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                       C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p]
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def time_pair(src, work, reps=6):
    for _ in range(2):
        st = lib.c21hip_split_filter_xy(src.data_ptr(), work.data_ptr(), n, n, n, 1.5 * n, 1.5 * n, 0, 10.0, 0.0, 0,
                                        stream)
        assert st == 0, st
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        lib.c21hip_split_filter_xy(src.data_ptr(), work.data_ptr(), n, n, n, 1.5 * n, 1.5 * n, 0, 10.0, 0.0, 0, stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


pairs = []
for t in range(trials):
    src = torch.randn(nf, device="cuda", dtype=torch.float32)
    work = torch.empty(nf, device="cuda", dtype=torch.float32)
    pairs.append((src, work))
    print(f"pair {t}: src {src.data_ptr():#x} work {work.data_ptr():#x}  X+Y {time_pair(src, work):.3f} ms", flush=True)
for t in range(min(3, trials)):
    print(f"pair {t} again: X+Y {time_pair(*pairs[t]):.3f} ms", flush=True)
# the same work buffer with another source, and the other way round
print(f"src 0 -> work 1: {time_pair(pairs[0][0], pairs[1][1]):.3f} ms;  src 1 -> work 0: {time_pair(pairs[1][0], pairs[0][1]):.3f} ms")
