"""Pass Y (kind 1) time against the placement of the workspace: release the device cache, put a
dummy allocation of varying size in front, time again."""
import ctypes as C, importlib, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
lib = importlib.import_module("21cmfast_amd").load(require_gpu=True)
lib.c21hip_bench_pass.restype = C.c_int
lib.c21hip_bench_pass.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_double, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
lib.c21hip_ws.restype = C.c_void_p
lib.c21hip_ws.argtypes = [C.c_int, C.c_size_t]
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
keep = []
def t(kind):
    ms = C.c_float()
    st = lib.c21hip_bench_pass(kind, 512, 0, 3, 12.0, 37.0, 768.0, 10, stream, C.byref(ms))
    return ms.value * 1e3 if st == 0 else -1
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
for trial in range(12):
    lib.c21cm_release_device_cache()
    pad = ctypes.c_void_p()
    size = (trial * 37 % 11) * (1 << 20) + (trial % 3) * 4096 + 256
    hip.hipMalloc(ctypes.byref(pad), ctypes.c_size_t(size))
    keep.append(pad)
    y = t(1); x = t(8); z = t(2)
    a = lib.c21hip_ws(90, 8); b = lib.c21hip_ws(91, 8)
    print(f"trial {trial:2d} pad {size:9d}  Y {y:7.1f}  Xpair {x:7.1f}  Z {z:7.1f}   slot90 {a:#x} slot91 {b:#x}", flush=True)
