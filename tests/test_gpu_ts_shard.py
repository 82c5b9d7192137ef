"""ComputeTsBox with the N_STEP_TS shells dealt over ranks (reference:
SpinTemperatureBox.c:1541-1784 is linear in the shells, so the per-cell sums of the ranks' shells
add up to the sequential loop's).  One GPU here: the two compute phases of every rank of a world of
2, 3 and 8 run one after the other through the exported C entry points, the exchange between them
(the sum over ranks, cell slabs) is done with torch on the device -- the RCCL transport itself is
exercised on a one-rank communicator (C21CM_SHARD_TS=force).

Tolerance: the sums are formed in double in a different order (per rank, then over ranks), the
outputs are floats: x_e and T_k agree to a float ulp or two, T_s like everywhere (fixed point that
stops at a 1e-3 step)."""

import ctypes as C
import importlib
from pathlib import Path

import numpy as np
import pytest

import ts_helpers as H

pytestmark = pytest.mark.gpu
S = importlib.import_module("21cmfast_amd.structs")
D = importlib.import_module("21cmfast_amd.distributed")
DATA = Path(__file__).parent / "golden" / "reference" / "_data"
FIELDS = ("spin_temperature", "kinetic_temp_neutral", "xray_ionised_fraction")


def setup(lib, tmp_path, n, source_model):
    import torch
    from test_gpu_abi import Session

    ses = Session(lib, tmp_path, data_dir=DATA, HII_DIM=n, DIM=2 * n, BOX_LEN=1.5 * n,
                  SOURCE_MODEL=source_model, USE_TS_FLUCT=True, USE_LYA_HEATING=False, Z_HEAT_MAX=30.0,
                  N_THREADS=4)
    lib.init_heat.restype = C.c_int
    assert lib.init_heat() == 0, lib.c21cm_last_error()
    rng = np.random.default_rng(11)
    shape = (n, n, n)
    density = H.smooth_field(shape, rng, 0.3)
    prev = {"xray_ionised_fraction": np.exp(rng.uniform(np.log(1.5e-4), np.log(4e-4), shape)).astype(np.float32),
            "kinetic_temp_neutral": (9.0 * (1 + 0.6 * density)).astype(np.float32),
            "spin_temperature": np.full(shape, 30.0, np.float32)}
    d = {"density": torch.from_numpy(density).cuda()}
    d.update({k: torch.from_numpy(v).cuda() for k, v in prev.items()})
    return ses, d


def fp(t):
    return C.cast(t.data_ptr(), S.c_float_p)


def declare(lib):
    f32 = C.c_float
    lib.ComputeTsBox.restype = C.c_int
    lib.ComputeTsBox.argtypes = [f32, f32, f32, C.c_short] + [C.c_void_p] * 5
    lib.c21cm_ts_box_shard_sums.restype = C.c_int
    lib.c21cm_ts_box_shard_sums.argtypes = [f32, f32, f32, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                            C.c_void_p, C.POINTER(C.c_int)]
    lib.c21cm_ts_box_shard_finish.restype = C.c_int
    lib.c21cm_ts_box_shard_finish.argtypes = [f32, f32, f32, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_size_t, C.c_size_t, C.c_void_p]
    lib.c21cm_ts_slab_begin.restype = C.c_size_t
    lib.c21cm_ts_slab_begin.argtypes = [C.c_size_t, C.c_int, C.c_int]


@pytest.mark.parametrize("source_model", [1, 0])  # E-INTEGRAL (ln SFRD tables), CONST-ION-EFF (dfcoll/dz)
def test_shard_phases_reproduce_the_single_pass(gpu_lib, tmp_path, source_model):
    import torch

    lib = gpu_lib
    declare(lib)
    n = 64
    ses, d = setup(lib, tmp_path, n, source_model)
    z, prev_z = 14.0, 14.3
    ntot = n**3
    pf = S.PerturbedFieldStruct(density=fp(d["density"]))
    prevs = S.TsBoxStruct(**{k: fp(d[k]) for k in FIELDS})
    one = {k: torch.zeros((n, n, n), dtype=torch.float32, device="cuda") for k in FIELDS}
    outs = S.TsBoxStruct(**{k: fp(v) for k, v in one.items()})
    assert lib.ComputeTsBox(z, prev_z, z, 0, C.byref(pf), None, C.byref(prevs), None, C.byref(outs)) == 0, \
        lib.c21cm_last_error()
    torch.cuda.synchronize()
    assert float(one["kinetic_temp_neutral"].std()) > 0 and bool(torch.isfinite(one["spin_temperature"]).all())
    for world in (2, 3, 8):
        rows = C.c_int(0)
        total = None
        for rank in range(world):  # rank order: the order of the combine kernel
            part = torch.full((6, ntot), float("nan"), dtype=torch.float64, device="cuda")
            st = lib.c21cm_ts_box_shard_sums(z, prev_z, z, C.byref(pf), C.byref(prevs), rank, world,
                                             C.c_void_p(part.data_ptr()), C.byref(rows))
            assert st == 0, lib.c21cm_last_error()
            assert rows.value == 4
            part = part[: rows.value]
            assert bool(torch.isfinite(part).all())
            total = part.clone() if total is None else total + part
        got = {k: torch.zeros((n, n, n), dtype=torch.float32, device="cuda") for k in FIELDS}
        gouts = S.TsBoxStruct(**{k: fp(v) for k, v in got.items()})
        covered = 0
        for rank in range(world):
            b, e = D.ts_slab(ntot, rank, world)
            assert (b, e) == (lib.c21cm_ts_slab_begin(ntot, world, rank), lib.c21cm_ts_slab_begin(ntot, world, rank + 1))
            slab = total[:, b:e].contiguous()
            st = lib.c21cm_ts_box_shard_finish(z, prev_z, z, C.byref(pf), C.byref(prevs),
                                               C.c_void_p(slab.data_ptr()), b, e - b, C.byref(gouts))
            assert st == 0, lib.c21cm_last_error()
            covered += e - b
        torch.cuda.synchronize()
        assert covered == ntot
        for k in ("xray_ionised_fraction", "kinetic_temp_neutral"):
            a, b_ = got[k].cpu().numpy(), one[k].cpu().numpy()
            np.testing.assert_allclose(a, b_, rtol=3e-7, atol=0, err_msg=f"world {world} {k}")
            assert np.mean(a == b_) > 0.9
        a, b_ = got["spin_temperature"].cpu().numpy(), one["spin_temperature"].cpu().numpy()
        np.testing.assert_allclose(a, b_, rtol=2e-3)
        assert np.mean(np.abs(a / b_ - 1) < 1e-6) > 0.99
        assert gouts.Q_HI == outs.Q_HI
    del ses


def test_shell_deal_and_slabs():
    for n_step in (1, 7, 40):
        for world in (1, 2, 3, 8):
            seen = sorted(sum((D.shells_of_rank(n_step, r, world) for r in range(world)), []))
            assert seen == list(range(n_step))
    for ntot in (64**3, 50**3, 24 * 24 * 20):
        for world in (1, 2, 3, 8):
            edges = [D.ts_slab(ntot, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == ntot
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            assert all(b % 4 == 0 for b, _ in edges)


def test_compute_ts_box_shards_itself_over_a_one_rank_communicator(gpu_lib, tmp_path, monkeypatch):
    """The RCCL path end to end on what a one-GPU box can run: communicator of one rank,
    C21CM_SHARD_TS=force -> c21cm_ts_box_sharded (phase 1, combine, phase 2); bit-identical outputs
    are not required (another summation grouping), float agreement is."""
    import torch

    api = importlib.import_module("21cmfast_amd.grid_api")
    lib = gpu_lib
    declare(lib)
    n = 64
    ses, d = setup(lib, tmp_path, n, 1)
    z, prev_z = 14.0, 14.3
    pf = S.PerturbedFieldStruct(density=fp(d["density"]))
    prevs = S.TsBoxStruct(**{k: fp(d[k]) for k in FIELDS})

    def run():
        o = {k: torch.zeros((n, n, n), dtype=torch.float32, device="cuda") for k in FIELDS}
        os_ = S.TsBoxStruct(**{k: fp(v) for k, v in o.items()})
        assert lib.ComputeTsBox(z, prev_z, z, 0, C.byref(pf), None, C.byref(prevs), None, C.byref(os_)) == 0, \
            lib.c21cm_last_error()
        torch.cuda.synchronize()
        return o

    one = run()
    api.shard_init_single()
    monkeypatch.setenv("C21CM_SHARD_TS", "force")
    lib.c21cm_ts_box_sharded_calls.restype = C.c_int
    before = lib.c21cm_ts_box_sharded_calls()
    try:
        got = run()
    finally:
        api.shard_finalize()
    # the call DID shard (the gate ignored `force` and the test compared the replicated path
    # with itself)
    assert lib.c21cm_ts_box_sharded_calls() == before + 1
    for k in FIELDS:  # one rank: the same shells, summed per rank in double and rounded once
        assert torch.allclose(got[k], one[k], rtol=3e-7, atol=0.0), k
        assert float((got[k] == one[k]).float().mean()) > 0.9, k
    del ses
