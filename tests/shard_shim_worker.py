"""One rank of tests/test_gpu_shard_shim.py: launched by torch.distributed.run, N processes on the ONE GPU of
the box, the library's communicator formed over tests/shim/librccl_shim.so (C21CM_RCCL_LIB).  Every exchange of
csrc/host/shard_rccl.c then runs with world = N for real: the slab all-to-all of the packed first crossings, the
all-gather of chunk sums / flag / output slabs inside the slab finish, the bit gather onto the owner, the
ncclReduce of uint8 grids and 64-bit keys, the two-hop (first crossing, Gamma_12) exchange of the fused
recombination loop, the output broadcasts, the status agreement, and ComputeTsBox's reduce-scatter / all-gather.

Each case compares this rank's sharded result with the single pass computed by the same process (bit for bit
for ComputeIonizedBox: "the largest radius that ionises the cell" is order independent, reference
IonisationBox.c:1531-1588, and the finish is per cell, :1031-1256,1597-1608).  Failures are collected, not
raised, so that every rank keeps entering the collectives; the result goes to <tmp>/result_rank<r>.json."""

import ctypes as C
import importlib
import json
import os
import sys
import traceback
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

pkg = importlib.import_module("21cmfast_amd")
W = importlib.import_module("21cmfast_amd.workloads")
S = importlib.import_module("21cmfast_amd.structs")
D = importlib.import_module("21cmfast_amd.distributed")
api = importlib.import_module("21cmfast_amd.grid_api")

FIELDS = ("neutral_fraction", "z_reion", "kinetic_temperature")
RC_FIELDS = FIELDS + ("ionisation_rate_G12", "mean_free_path", "cumulative_recombinations")


class Ctx:
    def __init__(self):
        dist.init_process_group("gloo")
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        torch.cuda.set_device(0)
        self.lib = pkg.load(require_gpu=True)
        self.shim = C.CDLL(os.environ["C21CM_RCCL_LIB"])
        self.failures, self.done, self.info = [], [], {}
        api.shard_init_from_torch()
        self.lib.c21cm_shard_is_rccl.restype = C.c_int
        self.lib.c21cm_shard_comm_count.restype = C.c_int
        self.lib.c21cm_shard_last_finish_was_slab.restype = C.c_int
        self.check("communicator", self.lib.c21cm_shard_is_rccl() == 1 and
                   self.lib.c21cm_shard_comm_count() == self.world)

    def check(self, what, ok, detail=""):
        if not ok:
            self.failures.append(f"{what} {detail}".strip())

    def stats(self):
        st = (C.c_uint64 * 6)()
        assert self.shim.rccl_shim_stats(st) == 0
        return dict(zip(("sends", "recvs", "groups", "bytes_sent", "bytes_recv", "collectives"), [int(x) for x in st]))

    def agree_float(self, what, value):
        """the same double on every rank (scalars every rank must end up with)"""
        vals = [None] * self.world
        dist.all_gather_object(vals, float(value))
        self.check(what, len(set(vals)) == 1, f"differs over ranks: {vals}")

    def compare(self, what, got, ref, names, lo=None, hi=None):
        for k in names:
            a, b = getattr(got, k), getattr(ref, k)
            if a is None or b is None:
                self.check(what, a is None and b is None, f"{k}: one side missing")
                continue
            if isinstance(a, np.ndarray):
                a, b = torch.from_numpy(a), torch.from_numpy(b) if isinstance(b, np.ndarray) else b.cpu()
            elif isinstance(b, np.ndarray):
                b = torch.from_numpy(b).to(a.device)
            a, b = a.reshape(-1), b.reshape(-1)
            if lo is not None and a.numel() > 1:
                a, b = a[lo:hi], b[lo:hi]
            if not torch.equal(a, b):
                bad = int((a != b).sum())
                self.check(what, False, f"{k}: {bad} of {a.numel()} cells differ (rank {self.rank})")


def env(**kw):
    class _E:
        def __enter__(self):
            self.old = {k: os.environ.get(k) for k in kw}
            for k, v in kw.items():
                os.environ[k] = v

        def __exit__(self, *a):
            for k, v in self.old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return _E()


def sharded(c, spec, density, mode, **kw):
    """c21cm_ionize_sharded with the output mode as an int (1 whole boxes, 2 x_HI whole, 0 slab / owner)."""
    buf, box, rep = api.ionize_sharded(spec, density, broadcast=mode, **kw)
    torch.cuda.synchronize()
    return buf, rep


def case_lagrangian(c):
    """G = 2 fused Lagrangian loop (the benchmark's model): slab finish in its three output modes, the owner
    finish with the bit gather and with ncclReduce(uint8), device and host arrays."""
    n = 64
    spec = W.ionize_spec(n, r_bubble_max=20.0)
    density = torch.from_numpy(W.density_field_numpy(n, seed=13)).cuda()
    kw = dict(n_ion=W.nion_from_density(density))
    ref, _, rep0 = api.ionize_grids(spec, density, **kw)
    ion = float((ref.neutral_fraction == 0).float().mean())
    c.check("lagrangian workload", 0.03 < ion < 0.97, f"ionised fraction {ion}")
    sl = api.shard_slab(spec, c.rank, c.world)
    lo, hi = sl["cell_begin"], sl["cell_end"]
    s0 = c.stats()
    for mode, what in ((1, "slab finish, whole boxes"), (2, "slab finish, x_HI whole"), (0, "slab finish, slab-resident")):
        buf, rep = sharded(c, spec, density, mode, **kw)
        c.check(what, c.lib.c21cm_shard_last_finish_was_slab() == 1, "did not take the slab finish")
        if mode == 1:
            c.compare(what, buf, ref, FIELDS)
        else:
            c.compare(what, buf, ref, FIELDS, lo, hi)
            if mode == 2:
                c.compare(what, buf, ref, FIELDS[:1])
        c.check(what, rep.global_xH == rep0.global_xH, f"global_xH {rep.global_xH} vs {rep0.global_xH}")
    s1 = c.stats()
    # three calls: per call one slab all-to-all + chunk sums (2 messages per peer) [+ output slabs]
    c.check("slab exchanges executed", s1["sends"] - s0["sends"] >= 3 * 3 * (c.world - 1) and
            s1["groups"] - s0["groups"] >= 7, str((s0, s1)))
    ph = (C.c_double * 3)()
    c.lib.c21cm_shard_last_phases.restype = C.c_int
    c.check("phase marks", c.lib.c21cm_shard_last_phases(ph) == 0 and all(x >= 0 for x in ph))
    c.info["phases_ms"] = [round(x, 3) for x in ph]
    with env(C21CM_SHARD_FINISH="owner"):
        for ex in ("gather", "reduce"):
            with env(C21CM_SHARD_EXCHANGE=ex):
                buf, rep = sharded(c, spec, density, 1, **kw)
                what = f"owner finish, {ex}, broadcast"
                c.check(what, c.lib.c21cm_shard_last_finish_was_slab() == 0)
                c.compare(what, buf, ref, FIELDS)
                c.check(what, rep.global_xH == rep0.global_xH)
        # no broadcast: only the owner holds the box
        buf, rep = sharded(c, spec, density, 0, **kw)
        if c.rank == D.owner_rank(spec.n_radii, c.world):
            c.compare("owner finish, no broadcast", buf, ref, FIELDS)
    # host arrays (the reference's caller): staged broadcasts / staged slabs
    hd = density.cpu().numpy()
    hkw = dict(n_ion=kw["n_ion"].cpu().numpy())
    for finish in ("slab", "owner"):
        with env(C21CM_SHARD_FINISH=finish):
            buf, rep = sharded(c, spec, hd, 1, **hkw)
            c.compare(f"host arrays, {finish} finish, whole boxes", buf, ref, FIELDS)
            c.check(f"host arrays, {finish} finish", rep.global_xH == rep0.global_xH)
    c.agree_float("global_xH over ranks", rep.global_xH)
    c.done.append("lagrangian")


def case_means(c):
    """A Lagrangian loop that stops above the cell-scale radius returns the grid mean of its LAST radius
    (IonisationBox.c:1623-1628): the per-radius means travel with ncclReduce(float64, sum)."""
    n = 64
    spec = W.ionize_spec(n, r_bubble_max=12.0, r_lowest=3)
    density = torch.from_numpy(W.density_field_numpy(n, seed=5)).cuda()
    kw = dict(n_ion=W.nion_from_density(density))
    ref, box0, rep0 = api.ionize_grids(spec, density, **kw)
    buf, box, rep = api.ionize_sharded(spec, density, broadcast=1, **kw)
    torch.cuda.synchronize()
    c.compare("early stop (means reduce)", buf, ref, FIELDS)
    c.check("early stop mean_f_coll", box.mean_f_coll == box0.mean_f_coll, f"{box.mean_f_coll} vs {box0.mean_f_coll}")
    c.done.append("means")


def case_eulerian(c):
    """Closed-form (erfc) Eulerian loop: the slab finish of the one-sweep cell-scale radius."""
    n = 64
    for mode in (W.FCOLL_ERFC,):
        spec = W.ionize_spec(n, mode=mode, r_bubble_max=12.0)
        density = torch.from_numpy(W.density_field_numpy(n, seed=21)).cuda()
        ref, _, rep0 = api.ionize_grids(spec, density)
        buf, rep = sharded(c, spec, density, 1)
        what = "eulerian erfc, whole boxes"
        c.check(what, c.lib.c21cm_shard_last_finish_was_slab() == (1 if api.shard_slab_supported(spec) else 0))
        c.compare(what, buf, ref, FIELDS)
        c.check(what, rep.global_xH == rep0.global_xH, f"{rep.global_xH} vs {rep0.global_xH}")
        ion = float((ref.neutral_fraction == 0).float().mean())
        c.check("eulerian workload", 0.02 < ion < 0.98, f"ionised fraction {ion}")
    c.done.append("eulerian")


def case_recomb(c):
    """Recombination models: the fused loop's 5 byte / cell two-hop exchange (exchange_cross_g12) where it is
    supported (256-point z-lines), the 64-bit key reduce otherwise and with C21CM_SHARD_EXCHANGE=keys."""
    from recomb_helpers import inputs, recomb_spec

    n = 256
    spec = recomb_spec(n, model=2, cell_recomb=1, r_bubble_max=20.0)
    c.check("fused recombination spec", api.shard_rc_supported(spec))
    d = {k: torch.from_numpy(v).cuda() for k, v in inputs((n, n, n), seed=9).items()}
    kw = dict(n_ion=d["n_ion"], whalo_sfr=d["whalo_sfr"], prev_nrec=d["prev_nrec"], prev_z_reion=d["prev_z_reion"])
    ref, _, rep0 = api.ionize_grids(spec, d["density"], **kw)
    keep = {k: getattr(ref, k).clone() for k in RC_FIELDS}
    s0 = c.stats()
    buf, rep = sharded(c, spec, d["density"], 1, **kw)
    s1 = c.stats()
    for k in RC_FIELDS:
        if not torch.equal(keep[k], getattr(buf, k)):
            c.check("fused recombination exchange", False, f"{k}: {int((keep[k] != getattr(buf, k)).sum())} cells differ")
    c.check("fused recombination exchange", rep.global_xH == rep0.global_xH)
    # hop 1: two messages to every peer; hop 2: two to the owner from everybody else
    c.check("fused recombination exchange ran", s1["sends"] - s0["sends"] >= 2 * (c.world - 1), str((s0, s1)))
    # a small box: not supported by the fused loop -> keys
    n = 64
    spec = recomb_spec(n, model=2, cell_recomb=0)
    d = {k: torch.from_numpy(v).cuda() for k, v in inputs((n, n, n), seed=8).items()}
    kw = dict(n_ion=d["n_ion"], whalo_sfr=d["whalo_sfr"], prev_nrec=d["prev_nrec"], prev_z_reion=d["prev_z_reion"])
    ref, _, rep0 = api.ionize_grids(spec, d["density"], **kw)
    buf, rep = sharded(c, spec, d["density"], 1, **kw)
    c.compare("key reduce", buf, ref, RC_FIELDS)
    c.check("key reduce", rep.global_xH == rep0.global_xH)
    c.done.append("recomb")


def case_failure(c):
    """One rank fails its local phase (no n_ion grid for a Lagrangian model): EVERY rank returns an error from
    the same call, nobody is left in a receive, and the communicator still works afterwards."""
    n = 64
    spec = W.ionize_spec(n, r_bubble_max=12.0)
    density = torch.from_numpy(W.density_field_numpy(n, seed=3)).cuda()
    n_ion = W.nion_from_density(density)
    ref, _, rep0 = api.ionize_grids(spec, density, n_ion=n_ion)
    bad = c.world - 1
    for finish in ("slab", "owner"):
        with env(C21CM_SHARD_FINISH=finish):
            try:
                api.ionize_sharded(spec, density, n_ion=None if c.rank == bad else n_ion, broadcast=1)
                c.check(f"forced failure ({finish})", False, "the call returned 0")
            except RuntimeError as e:
                msg = str(e)
                c.check(f"forced failure ({finish})", ("another rank failed" in msg) == (c.rank != bad), msg)
            buf, rep = sharded(c, spec, density, 1, n_ion=n_ion)
            c.compare(f"after a failure ({finish})", buf, ref, FIELDS)
    c.done.append("failure")


def case_abi(c, tmp):
    """ComputeIonizedBox itself on a communicator: the reference's caller (host arrays, no knowledge of slabs)
    gets WHOLE boxes on every rank by default."""
    from test_gpu_abi import Session, call_ionize

    lib = c.lib
    for model, over in ((0, dict(HII_FILTER=1, USE_EXP_FILTER=False, USE_INTERPOLATION_TABLES=2)),):
        ses = Session(lib, Path(tmp), HII_DIM=64, SOURCE_MODEL=model, R_BUBBLE_MAX=12.0, HII_EFF_FACTOR=60.0,
                      M_MIN_in_Mass=True, **over)
        density = W.density_field_numpy(64, seed=3, sigma=0.6)
        with env(C21CM_SHARD="0"):
            one = call_ionize(lib, 8.0, density, need_nion=True)
        got = call_ionize(lib, 8.0, density, need_nion=True)
        c.check("ABI sharded status", one["status"] == 0 and got["status"] == 0, str(lib.c21cm_last_error()))
        for k in FIELDS:
            if not np.array_equal(one[k], got[k]):
                bad = np.flatnonzero(one[k].reshape(-1) != got[k].reshape(-1))
                i0 = int(bad[0])
                c.check("ABI ComputeIonizedBox whole boxes", False,
                        f"{k}: {bad.size} cells differ, flat indices {i0}..{int(bad[-1])}, first: single "
                        f"{one[k].reshape(-1)[i0]!r} sharded {got[k].reshape(-1)[i0]!r} (rank {c.rank})")
        c.check("ABI mean_f_coll", one["mean_f_coll"] == got["mean_f_coll"])
        ion = float((one["neutral_fraction"] == 0).mean())
        c.check("ABI workload", 0.02 < ion < 0.98, f"{ion}")
        del ses
    c.done.append("abi")


def case_ts(c, tmp):
    """ComputeTsBox on the communicator (c21cm_ts_box_sharded): reduce-scatter of the shell sums by cell slabs,
    temperature update of the slab, all-gather of the three boxes.  The sums are formed in double per rank and
    then over ranks: x_e / T_k equal the single pass to a float ulp or two (tests/test_gpu_ts_shard.py), and
    every rank must hold the SAME boxes."""
    import zlib

    from test_gpu_abi import Session
    from test_gpu_ts_shard import FIELDS as TSF, declare, fp, setup

    lib = c.lib
    declare(lib)
    lib.c21cm_ts_box_sharded_calls.restype = C.c_int
    n = 64
    for source_model in (1, 0):
        ses, d = setup(lib, Path(tmp), n, source_model)
        z, prev_z = 14.0, 14.3
        pf = S.PerturbedFieldStruct(density=fp(d["density"]))
        prevs = S.TsBoxStruct(**{k: fp(d[k]) for k in TSF})

        def run():
            o = {k: torch.zeros((n, n, n), dtype=torch.float32, device="cuda") for k in TSF}
            os_ = S.TsBoxStruct(**{k: fp(v) for k, v in o.items()})
            st = lib.ComputeTsBox(z, prev_z, z, 0, C.byref(pf), None, C.byref(prevs), None, C.byref(os_))
            torch.cuda.synchronize()
            return st, o

        with env(C21CM_SHARD_TS="0"):
            st0, one = run()
        before = lib.c21cm_ts_box_sharded_calls()
        for exch in ("f64", "f32"):
            with env(C21CM_TS_SHARD_EXCHANGE=exch):
                st1, got = run()
            what = f"ComputeTsBox sharded ({'E-INTEGRAL' if source_model else 'CONST-ION-EFF'}, {exch})"
            c.check(what, st0 == 0 and st1 == 0, str(lib.c21cm_last_error()))
            rtol = 3e-7 if exch == "f64" else 2e-6
            for k in ("xray_ionised_fraction", "kinetic_temp_neutral"):
                c.check(what, torch.allclose(got[k], one[k], rtol=rtol, atol=0.0),
                        f"{k}: max rel {float(((got[k] - one[k]).abs() / one[k].abs()).max()):.3e}")
            c.check(what, torch.allclose(got["spin_temperature"], one["spin_temperature"], rtol=2e-3))
            crcs = [None] * c.world
            dist.all_gather_object(crcs, [zlib.crc32(got[k].cpu().numpy().tobytes()) for k in TSF])
            c.check(what + " all ranks hold the same boxes", all(x == crcs[0] for x in crcs), str(crcs))
        c.check("ComputeTsBox took the sharded path", lib.c21cm_ts_box_sharded_calls() == before + 2)
        del ses
    c.done.append("ts")


def main():
    cases = sys.argv[1].split(",")
    c = Ctx()
    # (a directory of its own per rank: the sessions of the ABI / TsBox cases write their synthetic RECFAST table there,
    #  and eight ranks writing and reading ONE file raced -- a rank loaded a half-written table)
    tmp = str(Path(sys.argv[2]) / f"rank{c.rank}")
    Path(tmp).mkdir(parents=True, exist_ok=True)
    table = {"lagrangian": case_lagrangian, "means": case_means, "eulerian": case_eulerian, "recomb": case_recomb,
             "failure": case_failure, "abi": lambda c_: case_abi(c_, tmp), "ts": lambda c_: case_ts(c_, tmp)}
    for name in cases:
        try:
            table[name](c)
        except Exception:
            c.failures.append(f"{name}: exception\n{traceback.format_exc()}")
            break  # the other ranks may be inside a collective this rank left: stop here (the shim times out)
    stats = c.stats()
    api.shard_finalize()
    # (a file per rank: eight ranks printing at once interleave their lines on the launcher's stdout)
    result = json.dumps({"rank": c.rank, "world": c.world, "failures": c.failures, "done": c.done, "stats": stats,
                         "info": c.info})
    Path(sys.argv[2], f"result_rank{c.rank}.json").write_text(result)
    print("RESULT " + result, flush=True)
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        pass
    sys.exit(1 if c.failures else 0)


if __name__ == "__main__":
    main()
