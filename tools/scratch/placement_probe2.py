"""Two-grid passes X + Y (no window) on several buffer quadruples inside one process: do the two speeds of the
1024^3 line passes belong to the buffers?  GPU box only."""
import ctypes as C
import importlib
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lib = importlib.import_module("21cmfast_amd").load(require_gpu=True)
lib.c21hip_split_floats.restype = C.c_size_t
lib.c21hip_split_floats.argtypes = [C.c_int] * 3
nf = lib.c21hip_split_floats(n, n, n)
lib.c21hip_split_filter_xy2.restype = C.c_int
lib.c21hip_split_filter_xy2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_float, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_float,
                                        C.c_int, C.c_int, C.c_int, C.c_void_p]
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(q):
    st = lib.c21hip_split_filter_xy2(q[0].data_ptr(), q[1].data_ptr(), 0, 0.0, q[2].data_ptr(), q[3].data_ptr(), 0, 0.0,
                                     n, n, n, 1.5 * n, 1.5 * n, 10.0, 0, 0, 1, stream)
    assert st == 0, st


def time_quad(q, reps=5):
    run(q), run(q)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        run(q)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


quads = []
for t in range(trials):
    q = [torch.randn(nf, device="cuda", dtype=torch.float32) if i % 2 == 0 else
         torch.empty(nf, device="cuda", dtype=torch.float32) for i in range(4)]
    quads.append(q)
    print(f"set {t}: " + " ".join(f"{x.data_ptr():#x}" for x in q) + f"  X+Y (2 grids) {time_quad(q):.3f} ms", flush=True)
for t in range(min(3, trials)):
    print(f"set {t} again: {time_quad(quads[t]):.3f} ms", flush=True)
mix = [quads[0][0], quads[0][1], quads[1][2], quads[1][3]]
print(f"grid a of set 0 with grid b of set 1: {time_quad(mix):.3f} ms")

# ---- which buffers make a slow set slow?  (first slow set S against the first fast set F)
times = [time_quad(q, 3) for q in quads]
lo = min(times)
slow = [i for i, t in enumerate(times) if t > 1.08 * lo]
fast = [i for i, t in enumerate(times) if t <= 1.03 * lo]
if slow and fast:
    S, F = quads[slow[0]], quads[fast[0]]
    print(f"slow set {slow[0]} ({times[slow[0]]:.3f}) against fast set {fast[0]} ({times[fast[0]]:.3f})")
    combos = {"S.a + F.b": [S[0], S[1], F[2], F[3]], "F.a + S.b": [F[0], F[1], S[2], S[3]],
              "S sources, F works": [S[0], F[1], S[2], F[3]], "F sources, S works": [F[0], S[1], F[2], S[3]],
              "S.work_a only": [F[0], S[1], F[2], F[3]], "S.work_b only": [F[0], F[1], F[2], S[3]],
              "S.src_a only": [S[0], F[1], F[2], F[3]], "S.src_b only": [F[0], F[1], S[2], F[3]],
              "S with a and b swapped": [S[2], S[3], S[0], S[1]]}
    for name, q in combos.items():
        print(f"  {name:24s} {time_quad(q, 3):.3f} ms")
    # a slow PAIR of work buffers: grid b's work buffer shifted by multiples of 2 MB inside a larger allocation
    slack = 48 << 20  # floats
    big_w = torch.empty(nf + slack, device="cuda", dtype=torch.float32)
    wa = S[1]
    res = []
    for k in list(range(0, 17)) + [24, 32, 48, 64, 96]:
        off = k * (1 << 19)  # floats: k x 2 MB
        if off > slack:
            break
        q = [S[0], wa, S[2], big_w[off:off + nf]]
        res.append((k, time_quad(q, 3)))
    print("  S.work_a with a fresh work b shifted by k x 2 MB: " + " ".join(f"{k}:{t:.2f}" for k, t in res))
    big_w2 = torch.empty(nf + slack, device="cuda", dtype=torch.float32)
    res = []
    for k in list(range(0, 17)) + [24, 32, 48, 64, 96]:
        off = k * (1 << 19)
        q = [S[0], big_w2[off:off + nf], S[2], S[3]]
        res.append((k, time_quad(q, 3)))
    print("  a fresh work a shifted by k x 2 MB with S.work_b:  " + " ".join(f"{k}:{t:.2f}" for k, t in res))
else:
    print("no slow / fast pair among these sets", [round(t, 3) for t in times])
