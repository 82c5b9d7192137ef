"""The C-level multi-rank exchanges of csrc/host/shard_rccl.c EXECUTED with world = 2 and 3.

The boxes here have one MI355X and RCCL refuses two ranks on one device, so until round 6 every
ncclSend / ncclRecv / ncclGroupEnd of the library had only ever run on a one-rank communicator, where the peer
loops are empty.  Here N real processes (torch.distributed.run, as the driver launches bench.py) share the one
GPU and the library binds tests/shim/librccl_shim.so instead of librccl (C21CM_RCCL_LIB): shared-memory
channels with RCCL's group and rendezvous semantics, checked on their own in tests/test_rccl_shim.py.  Only the
transport differs from the 8-GPU run; xGMI timings remain unmeasured.

tests/shard_shim_worker.py holds the cases; every rank compares with the single pass it computes itself."""

import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
SHIM = ROOT / "tests" / "shim" / "librccl_shim.so"


def launch(world, cases, tmp_path, timeout=1500):
    if not SHIM.exists():
        subprocess.run(["make", "-C", str(SHIM.parent)], check=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # (256 KB slots: every message of the exchanges travels in several chunks, and eight ranks need 16 MB of /dev/shm)
    env = dict(os.environ, C21CM_RCCL_LIB=str(SHIM), RCCL_SHIM_TIMEOUT_S="120", RCCL_SHIM_SLOT_KB="256",
               C21CM_WS_PLACE="0", OMP_NUM_THREADS="4")
    env.pop("C21CM_SHARD", None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        str(ROOT / "tests" / "shard_shim_worker.py"), ",".join(cases), str(tmp_path)],
                       capture_output=True, text=True, timeout=timeout, env=env)
    results = [json.loads(f.read_text()) for f in sorted(Path(tmp_path).glob("result_rank*.json"))]
    return p, sorted(results, key=lambda r: r["rank"])


def check(p, results, world, cases):
    assert len(results) == world, (p.stdout[-3000:], p.stderr[-3000:])
    for r in results:
        assert not r["failures"], "\n".join(r["failures"])
        assert r["done"] == list(cases)
        assert r["stats"]["sends"] > 0 and r["stats"]["recvs"] > 0 and r["stats"]["groups"] > 0
    assert p.returncode == 0, p.stderr[-3000:]


@pytest.mark.parametrize("world", [2, 3, 8])  # 8: the node the budgets of DESIGN section 6 are written for
def test_ionized_box_exchanges_with_real_ranks(world, tmp_path):
    """Slab finish (three output modes), owner finish (bit gather, ncclReduce), host arrays, the per-radius
    means, the Eulerian slab finish, a forced failure on one rank, and ComputeIonizedBox through the ABI."""
    cases = ("lagrangian", "means", "eulerian", "failure", "abi")
    p, results = launch(world, cases, tmp_path)
    check(p, results, world, cases)


@pytest.mark.parametrize("world", [2, 3, 5])
def test_recombination_exchanges_with_real_ranks(world, tmp_path):
    """exchange_cross_g12 (5 bytes per cell, two hops) and the 64-bit key reduce."""
    cases = ("recomb",)
    p, results = launch(world, cases, tmp_path)
    check(p, results, world, cases)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_ts_box_exchanges_with_real_ranks(world, tmp_path):
    """c21cm_ts_box_sharded: reduce-scatter of the shell sums, all-gather of the boxes."""
    cases = ("ts",)
    p, results = launch(world, cases, tmp_path)
    check(p, results, world, cases)


@pytest.mark.parametrize("world", [2, 3])
def test_bench_ranks_through_the_c_exchange(world):
    """bench.py launched the way the driver launches it, N ranks on the one GPU: with the stand-in transport the
    sharded step runs INSIDE the C library (c21cm_ionize_sharded) -- the line carries one phase triple per rank
    and the communicator's own rank count -- and global x_HI equals the single-GPU run's."""
    from test_gpu_ionize import check_sharded_bench_objects

    common = ["--hii-dim", "128", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-roofline",
              "--no-abi", "--config4-dim", "256"]
    env = dict(os.environ, C21CM_WS_PLACE="0")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py")] + common, capture_output=True, text=True,
                       timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    single = json.loads(p.stdout.strip())
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env.update(C21CM_RCCL_LIB=str(SHIM), RCCL_SHIM_TIMEOUT_S="90")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"),
                        "--gpus", str(world), "--backend", "gloo"] + common,
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    cfg = line["config"]
    assert line["n_gpus"] == world and "c21cm_ionize_sharded" in cfg["parallelism"] and "shard_impl_note" not in cfg
    assert len(cfg["shard_phases_ms_per_rank"]) == world and cfg["rccl_comm_count"] == world
    assert all(len(t) == 3 and all(v >= 0 for v in t) for t in cfg["shard_phases_ms_per_rank"])
    assert cfg["global_xH"] == single["config"]["global_xH"]
    assert cfg["shard_outputs"].startswith("whole boxes") and cfg["ms_per_step_slab_resident"] > 0
    check_sharded_bench_objects(line, world=world, config4_dim=256)
    assert len(line["config4"]["shard_phases_ms_per_rank"]) == world
