"""Is the class of a work spectrum a function of its offset inside ONE large allocation?  A 72 GB arena; the
partner at offset 0, candidates at offsets k x step; two-grid pass Y on (partner, candidate).  GPU box only."""
import ctypes as C
import importlib
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

n = 512
lib = importlib.import_module("21cmfast_amd").load(require_gpu=True)
lib.c21hip_split_floats.restype = C.c_size_t
lib.c21hip_split_floats.argtypes = [C.c_int] * 3
nf = lib.c21hip_split_floats(n, n, n)
lib.c21hip_probe_pass_y2.restype = C.c_int
lib.c21hip_probe_pass_y2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
arena_gb = int(sys.argv[1]) if len(sys.argv) > 1 else 72
arena = torch.zeros(arena_gb << 28, device="cuda", dtype=torch.float32)  # floats: arena_gb GiB
print(f"arena {arena.data_ptr():#x}, {arena_gb} GiB")


def y2(off_a, off_b):
    ms = C.c_float()
    a, b = arena[off_a:off_a + nf], arena[off_b:off_b + nf]
    assert lib.c21hip_probe_pass_y2(a.data_ptr(), b.data_ptr(), n, n, n, 4, C.byref(ms), stream) == 0
    return ms.value * 1e3


G = 1 << 28  # floats per GiB
for step_name, offs in (("1 GiB steps", [k * G for k in range(1, arena_gb - 1)]),
                        ("64 MiB steps from 1 GiB", [G + k * (G // 16) for k in range(0, 33)])):
    ts = [y2(0, o) for o in offs]
    lo = min(ts)
    print(step_name + ": " + " ".join(f"{t / lo:4.2f}" for t in ts))
# partner in the middle
mid = (arena_gb // 2) * G
ts = [y2(mid, k * G) for k in range(0, arena_gb - 1) if abs(k * G - mid) >= G]
lo = min(ts)
print(f"partner at {arena_gb // 2} GiB, 1 GiB steps: " + " ".join(f"{t / lo:4.2f}" for t in ts))
