"""time selected c21hip_bench_pass kinds: python tools/scratch/tp_kinds.py N kind [kind ...]"""
import ctypes as C, importlib, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
n = int(sys.argv[1])
lib = importlib.import_module("21cmfast_amd").load(require_gpu=True)
lib.c21hip_bench_pass.restype = C.c_int
lib.c21hip_bench_pass.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_double, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
lib.c21cm_last_error.restype = C.c_char_p
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for kind in map(int, sys.argv[2:]):
    ms = C.c_float()
    st = lib.c21hip_bench_pass(kind, n, 0, 3, 12.0, 37.0, 1.5 * n, 10, stream, C.byref(ms))
    print(f"n={n} kind {kind} st={st} {ms.value*1e3:9.1f} us", lib.c21cm_last_error() if st else "")
