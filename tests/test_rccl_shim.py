"""The test-only librccl stand-in (tests/shim/rccl_shim.c) checked on its own, on the host (RCCL_SHIM_HOST=1:
memcpy instead of hipMemcpy), in real processes: group semantics, message matching, the collectives
shard_rccl.c uses, and that a mismatched exchange FAILS (size mismatch, unmatched send) instead of passing
or hanging.  The GPU tests (tests/test_gpu_shard_shim.py) then trust it to execute the library's multi-rank
exchanges."""

import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

import pytest

SHIM = Path(__file__).resolve().parent / "shim" / "librccl_shim.so"

WORKER = r"""
import ctypes as C, sys, numpy as np
shim = C.CDLL(sys.argv[1]); rank, world, case = int(sys.argv[2]), int(sys.argv[3]), sys.argv[5]
class Id(C.Structure): _fields_ = [("b", C.c_char * 128)]
uid = Id(); C.memmove(C.byref(uid), bytes.fromhex(sys.argv[4]), 128)
comm = C.c_void_p()
shim.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Id, C.c_int]
assert shim.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0
shim.ncclGetErrorString.restype = C.c_char_p
P = C.c_void_p
shim.ncclSend.argtypes = [P, C.c_size_t, C.c_int, C.c_int, P, P]
shim.ncclRecv.argtypes = [P, C.c_size_t, C.c_int, C.c_int, P, P]
shim.ncclAllReduce.argtypes = [P, P, C.c_size_t, C.c_int, C.c_int, P, P]
shim.ncclReduce.argtypes = [P, P, C.c_size_t, C.c_int, C.c_int, C.c_int, P, P]
shim.ncclBroadcast.argtypes = [P, P, C.c_size_t, C.c_int, C.c_int, P, P]
U8, I32, U64, F64, SUM, MAX = 1, 2, 5, 8, 0, 2
def ptr(a): return a.ctypes.data_as(P)
def size(src, dst, k): return 1000 * (1 + src) + 37 * dst + 5000 * k   # > one 1 KB slot, all different
def payload(src, dst, k): return ((np.arange(size(src, dst, k)) * (src + 3) + dst + 11 * k) % 251).astype(np.uint8)
if case == "alltoall":
    # every rank: two messages to every peer and two from every peer inside ONE group, sends posted first
    recv = {(p, k): np.zeros(size(p, rank, k), np.uint8) for p in range(world) if p != rank for k in (0, 1)}
    send = {(p, k): payload(rank, p, k) for p in range(world) if p != rank for k in (0, 1)}
    assert shim.ncclGroupStart() == 0
    for (p, k), a in sorted(send.items()): assert shim.ncclSend(ptr(a), a.size, U8, p, comm, None) == 0
    for (p, k), a in sorted(recv.items()): assert shim.ncclRecv(ptr(a), a.size, U8, p, comm, None) == 0
    rc = shim.ncclGroupEnd(); assert rc == 0, shim.ncclGetErrorString(rc)
    for (p, k), a in recv.items(): assert np.array_equal(a, payload(p, rank, k)), (p, k)
    # collectives
    a = np.array([rank * 7 % 5, -rank], np.int32)
    assert shim.ncclAllReduce(ptr(a), ptr(a), 2, I32, MAX, comm, None) == 0
    assert a.tolist() == [max(r * 7 % 5 for r in range(world)), 0]
    d = np.full(300000, 0.1 * (rank + 1)); acc = np.zeros(300000)
    assert shim.ncclReduce(ptr(d), ptr(acc if rank == 1 else d), d.size, F64, SUM, 1, comm, None) == 0
    if rank == 1:
        ref = np.zeros(300000)
        for r in range(world): ref = (ref + 0.1 * (r + 1)) if r else np.full(300000, 0.1)
        assert np.array_equal(acc, ref)
    k = np.full(1000, rank + 1, np.uint64) << np.uint64(40)
    assert shim.ncclReduce(ptr(k), ptr(k), k.size, U64, MAX, 0, comm, None) == 0
    if rank == 0: assert int(k[0]) == world << 40
    b = np.full(5000, rank, np.uint8)
    assert shim.ncclBroadcast(ptr(b), ptr(b), b.size, U8, world - 1, comm, None) == 0
    assert (b == world - 1).all()
    stats = (C.c_uint64 * 6)(); assert shim.rccl_shim_stats(stats) == 0
    assert stats[0] == 2 * (world - 1) and stats[1] == 2 * (world - 1) and stats[2] == 1
    print("OK")
elif case == "mismatch":
    # rank 0 sends 2000 bytes, rank 1 expects 1999: the receiver must see an error, the sender a timeout
    a = np.zeros(2000, np.uint8)
    rc = shim.ncclSend(ptr(a), 2000, U8, 1, comm, None) if rank == 0 else shim.ncclRecv(ptr(a), 1999, U8, 0, comm, None)
    print("RC", rc, shim.ncclGetErrorString(rc).decode())
elif case == "unmatched":
    # both ranks send first OUTSIDE a group: rendezvous semantics -> nobody receives -> both time out
    a = np.zeros(10, np.uint8)
    rc = shim.ncclSend(ptr(a), 10, U8, 1 - rank, comm, None)
    print("RC", rc, shim.ncclGetErrorString(rc).decode())
elif case == "selfsend":
    a = np.zeros(10, np.uint8)
    print("RC", shim.ncclSend(ptr(a), 10, U8, rank, comm, None), shim.ncclRecv(ptr(a), 10, U8, world, comm, None))
assert shim.ncclCommDestroy(comm) == 0
"""


def launch(tmp_path, world, case, timeout_s="120"):
    if not SHIM.exists():
        subprocess.run(["make", "-C", str(SHIM.parent)], check=True)
    shim = C.CDLL(str(SHIM))
    uid = (C.c_char * 128)()
    assert shim.ncclGetUniqueId(uid) == 0
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, RCCL_SHIM_HOST="1", RCCL_SHIM_SLOT_KB="1", RCCL_SHIM_TIMEOUT_S=timeout_s)
    procs = [subprocess.Popen([sys.executable, str(script), str(SHIM), str(r), str(world), bytes(uid).hex(), case],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(world)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-3000:]
    return [o for o, _ in outs]


@pytest.mark.parametrize("world", [2, 3, 5])
def test_grouped_all_to_all_and_collectives(tmp_path, world):
    assert all(o.strip().endswith("OK") for o in launch(tmp_path, world, "alltoall"))


def test_size_mismatch_is_an_error_not_a_pass(tmp_path):
    o = launch(tmp_path, 2, "mismatch", timeout_s="3")
    assert o[1].startswith("RC 4") and "receive of 1999 bytes met a send of 2000" in o[1]  # ncclInvalidArgument
    assert o[0].startswith("RC 2") and "timed out" in o[0]                                 # ncclSystemError


def test_unmatched_sends_outside_a_group_time_out(tmp_path):
    o = launch(tmp_path, 2, "unmatched", timeout_s="2")
    assert all(x.startswith("RC 2") and "timed out" in x for x in o)


def test_bad_peers_are_rejected(tmp_path):
    o = launch(tmp_path, 2, "selfsend")
    assert all(x.strip() == "RC 4 4" for x in o)
