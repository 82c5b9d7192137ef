#!/usr/bin/env python
"""bench.py -- IonizeBox cells/sec on MI355X (BASELINE.json metric).

A "step" is ONE ComputeIonizedBox grid pass (c21cm_ionize_grids through the C ABI) over
the synthetic config-3 workload of SURVEY.md 8(d): HII_DIM = 512, BOX_LEN = 768 Mpc,
40 filter radii (0.9305 .. 38.29 Mpc), G = 2 filtered grids (delta with the real-space
top-hat, n_ion with the exponential-MFP top-hat), first-snapshot path, inputs already
resident in HBM when the timed region starts.

  python bench.py                      # N = 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

N > 1 shards the R loop over ranks (radius index r belongs to rank (n_radii-1-r) % N),
max-reduces the per-cell first-crossing mask (uint8) onto the rank that owns radius 0
with one RCCL reduce over xGMI, and that rank finishes (partial ionisation, temperatures,
sum xH).  Total work is fixed as N grows -> "scaling": "strong".

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      algorithmic bytes / measured time of the dominant unit, vs 8 TB/s HBM3E
  cpu_baseline  the CPU oracle (oracle/liboracle21.so, "port") timed on this host on a
                bounded sample of the same workload (smaller box, same 40 radii)
"""

from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--hii-dim", type=int, default=512)
    ap.add_argument("--r-bubble-max", type=float, default=40.0)
    ap.add_argument("--mode", choices=["stars", "erfc", "icpf"], default="stars",
                    help="stars: the headline (ComputeIonizedBox, G = 2); erfc: its G = 1 closed form; "
                         "icpf: BASELINE config 2 -- InitialConditions + PerturbedField at "
                         "HII_DIM = --hii-dim / 2 (default 256), DIM = 2 HII_DIM, one GPU")
    ap.add_argument("--achievable-gbs", type=float, default=6200.0,
                    help="what a plain float4 copy kernel moves on these boxes (tools/copy_bench.hip, "
                         "profiles/r05_copy_bench.txt): printed next to the 8 TB/s peak in `roofline`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-dim", type=int, default=192,
                    help="box size of the CPU-oracle samples on hosts with < 16 cores (256 otherwise)")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo (host-staged) lets several ranks share "
                         "one GPU in the plumbing test, nccl = RCCL over xGMI is the real one")
    ap.add_argument("--shard-impl", default="c", choices=["c", "torch"],
                    help="where the mask reduce happens: c = inside the C library through its own "
                         "RCCL communicator (c21cm_ionize_sharded, the product path; nccl backend "
                         "only), torch = torch.distributed.reduce between the two C phases")
    ap.add_argument("--config4-dim", type=int, default=1024,
                    help="sharded runs (N > 1 or --force-shard) also time BASELINE config 4 -- this "
                         "HII_DIM x 40 radii sharded over the ranks, with its own same-run single-GPU "
                         "time -- and report it as the `config4` object; 0 skips it")
    ap.add_argument("--config4-steps", type=int, default=3)
    ap.add_argument("--no-abi", action="store_true",
                    help="skip the ComputeIonizedBox-through-the-ABI timings (abi_ms, h2d_d2h_ms)")
    ap.add_argument("--cpu-metric-box", action="store_true", default=None,
                    help="time the threaded CPU oracle on the metric's own box (HII_DIM^3, ~35 s at 512^3 "
                         "on 64 cores) -- default on hosts with >= 32 cores")
    ap.add_argument("--force-shard", action="store_true",
                    help="run the sharded code path (shard phase, RCCL reduce, finish phase) even "
                         "with one rank: a smoke test of the multi-GPU plumbing, not a benchmark")
    return ap.parse_args()


def algorithmic_bytes_per_radius(n_cells: int, G: int) -> float:
    """B_alg(R) = G*5S + 8N ~ (20G + 8) N bytes, S = one padded grid."""
    return (20.0 * G + 8.0) * n_cells


def parity_object(got, ref, n_radii):
    """HIP outputs against the CPU oracle's on the SAME inputs (reference
    IonisationBox.c:773-1201): fraction of cells whose ionisation flag differs, x_HI on the agreeing cells
    against rtol 1e-4 / atol 5e-6 (the north star's tolerance; the atol is the float32-transform round-off of
    1 - f zeta near the barrier), z_reion there exactly, per-radius f_coll grid means to rtol 1e-5.
    `got` / `ref`: dicts with neutral_fraction, z_reion (numpy) and report."""
    import numpy as np

    xg, xr = got["neutral_fraction"], ref["neutral_fraction"]
    ion_g, ion_r = xg == 0, xr == 0
    same = ion_g == ion_r
    mismatch = float(np.mean(~same))
    d = np.abs(xg[same].astype(np.float64) - xr[same])
    bound = 5e-6 + 1e-4 * np.abs(xr[same])
    mg = np.array(got["report"].f_coll_grid_mean[:n_radii], dtype=np.float64)
    mr = np.array(ref["report"].f_coll_grid_mean[:n_radii], dtype=np.float64)
    nz = mr != 0
    mean_rel = float(np.max(np.abs(mg[nz] / mr[nz] - 1.0))) if nz.any() else 0.0
    zre_equal = bool(np.array_equal(got["z_reion"][same], ref["z_reion"][same]))
    out = {
        "flag_mismatch": mismatch, "max_abs_dxH": float(d.max()) if d.size else 0.0,
        "xH_outside_tolerance": int(np.count_nonzero(d > bound)),
        "z_reion_equal_on_agreeing_cells": zre_equal, "fcoll_mean_max_rel": mean_rel,
        "d_global_xH": float(got["report"].global_xH - ref["report"].global_xH),
        "ionised_fraction": float(ion_r.mean()),
        "tolerance": "flag mismatch <= 2e-4; x_HI rtol 1e-4 + atol 5e-6 and z_reion equal on agreeing cells; "
                     "per-radius f_coll means rtol 1e-5",
    }
    out["pass"] = bool(mismatch <= 2e-4 and out["xH_outside_tolerance"] == 0 and zre_equal and mean_rel <= 1e-5)
    return out


def cpu_baseline(args, W, gpu_run=None):
    """Time the CPU oracle ("port": the reference's loop structure in C + OpenMP with its own FFT)
    on bounded samples of the same workload, three ways:
      threaded      N_THREADS = all cores (<= 64), transforms threaded too  -- charitable
      faithful_fft  N_THREADS = all cores, transforms single-threaded: what the reference's
                    dft.c effectively does (plans created per call, FFTW_ESTIMATE, :83-85)
      one_thread    N_THREADS = 1
    Each sample keeps the 40-radius ladder of the full workload on a smaller box (1.5 Mpc cells
    scaled so that the radii are the same multiples of the box); the slow variants run a subset
    of the radii and are scaled to the full ladder (stated in `sample`).  The headline `value`
    is the FASTEST variant, so gpu_over_cpu is the conservative ratio.
    `gpu_run(spec)` (round 6): returns the host copies of the GPU's input fields for this spec and the HIP
    outputs; the full-ladder variant then runs the oracle on THOSE fields and its result is compared with the
    HIP one -- the oracle run the line pays for is also the full-size parity check (`parity`)."""
    import numpy as np

    oracle = importlib.import_module("oracle.oracle")
    oracle.load()
    mode = W.FCOLL_STARS if args.mode == "stars" else W.FCOLL_ERFC
    full = W.ionize_spec(args.hii_dim, mode=mode, r_bubble_max=args.r_bubble_max)
    all_cores = min(os.cpu_count() or 1, 64)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass

    parity = {}

    def run(n, threads, fft_threads, n_keep, check=False):
        spec = W.ionize_spec(n, box_len=1.5 * args.hii_dim, mode=mode,
                             r_bubble_max=args.r_bubble_max)
        # every n_radii/n_keep-th radius of the full ladder (always the cell-scale index 0)
        idx = sorted({int(round(i * (full.n_radii - 1) / max(1, n_keep - 1))) for i in range(n_keep)})
        spec.n_radii = len(idx)
        for j, i in enumerate(idx):
            spec.R[j] = full.R[i]
            spec.sigma_maxmass[j] = full.sigma_maxmass[i]
        got = None
        if check and gpu_run is not None and len(idx) == full.n_radii:
            got = gpu_run(spec)  # the SAME realisation on both sides
            density, n_ion = got["density"], got["n_ion"]
        else:
            density = W.density_field_numpy(n, seed=12345)
            n_ion = W.nion_from_density(density) if mode == W.FCOLL_STARS else None
        oracle.set_threads(threads)
        oracle.set_fft_threads(fft_threads)
        t0 = time.perf_counter()
        ref = oracle.ionize_grids(spec, density, n_ion, need_nion=mode != W.FCOLL_STARS)
        dt = time.perf_counter() - t0
        oracle.set_fft_threads(0)
        if got is not None:
            parity.update(parity_object(got, ref, spec.n_radii))
            parity["box"] = n
            parity["what"] = (f"HIP pass vs the CPU oracle on the same {n}^3 density"
                              f"{' / n_ion' if n_ion is not None else ''} fields, {spec.n_radii} radii")
        del ref
        scale = full.n_radii / len(idx)  # time of the full ladder ~ per-radius time x 40
        return {"value": n**3 / (dt * scale), "seconds": dt, "box": n, "radii_run": len(idx),
                "threads": threads, "fft_threads": fft_threads or threads}

    G = 2 if mode == W.FCOLL_STARS else 1
    big = max(args.cpu_dim, 256) if all_cores >= 16 else args.cpu_dim
    # the threaded variant on the METRIC's box (VERDICT r3 weak point 8): HII_DIM^3 with the full
    # ladder, ~35 s at 512^3 on 64 threads; smaller hosts keep the scaled 256^3 / 192^3 sample
    metric_box = args.cpu_metric_box if args.cpu_metric_box is not None else (all_cores >= 32)
    variants = {
        "threaded": run(args.hii_dim if metric_box else big, all_cores, 0, full.n_radii, check=True),
        "faithful_fft": run(big, all_cores, 1, 12),
        "one_thread": run(128, 1, 1, full.n_radii),
    }
    best = max(variants, key=lambda k: variants[k]["value"])
    v = variants[best]
    return {
        "value": v["value"], "unit": "cells/s", "cores": v["threads"], "kind": "port",
        "sample": f"CPU oracle (C/OpenMP restatement of the reference loop, own FFT), variant "
                  f"'{best}': {v['box']}^3 box"
                  f"{' (the box of the metric)' if v['box'] == args.hii_dim else ''}, "
                  f"{v['radii_run']} of {full.n_radii} radii run "
                  f"(scaled to the full ladder), G={G}, {v['threads']} threads, "
                  f"{v['seconds']:.2f} s",
        "cpu_model": model, "host_cores": os.cpu_count(),
        "variants": variants,
        **({"parity": parity} if parity else {}),
    }


def abi_timing(n, args, torch, pkg, W):
    """One ComputeIonizedBox call of the benchmark workload THROUGH THE REFERENCE'S ENTRY POINT
    (parameter structs broadcast like py21cmfast does, SOURCE_MODEL = L-INTEGRAL: HaloBox.n_ion +
    PerturbedField.density in, three boxes out), once with device-resident (torch) arrays and once
    with host (numpy) arrays, which the library stages over PCIe: 2 inputs + the previous z_reion in,
    3 outputs back.  h2d_d2h_ms = the difference (reported separately, never in
    `value`)."""
    import ctypes as C

    import numpy as np

    S = importlib.import_module("21cmfast_amd.structs")
    lib = pkg.load()
    so = S.default_simulation_options(HII_DIM=n, DIM=2 * n, BOX_LEN=1.5 * n)
    mo = S.default_matter_options(SOURCE_MODEL=2)
    cp, ao, ct = S.default_cosmo_params(), S.default_astro_options(), S.default_cosmo_tables()
    ap = S.default_astro_params(R_BUBBLE_MAX=args.r_bubble_max)
    lib.Broadcast_struct_global_all(C.byref(so), C.byref(mo), C.byref(cp), C.byref(ap), C.byref(ao),
                                    C.byref(ct))
    data = ROOT / "tests" / "golden" / "reference" / "_data"
    path = str(data).encode()
    S.ConfigSettings.in_dll(lib, "config_settings").external_table_path = path
    lib.init_ps()
    z = 9.0
    density = W.density_field_torch(n, seed=12345)
    n_ion = W.nion_from_density(density)
    f32p = C.POINTER(C.c_float)

    def ptr(a):
        return C.cast(a.data_ptr(), f32p) if hasattr(a, "data_ptr") else a.ctypes.data_as(f32p)

    def run(arrs, reps):
        dens, nion, xh, zre, tk, prevz = arrs
        best = None
        for _ in range(reps):
            pf = S.PerturbedFieldStruct(density=ptr(dens))
            prev = S.IonizedBoxStruct(z_reion=ptr(prevz))
            ts, hb = S.TsBoxStruct(), S.HaloBoxStruct(n_ion=ptr(nion), log10_Mcrit_ACG_ave=8.7)
            box = S.IonizedBoxStruct(neutral_fraction=ptr(xh), z_reion=ptr(zre), kinetic_temperature=ptr(tk))
            ics = S.InitialConditionsStruct()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = lib.ComputeIonizedBox(z, 0.0, C.byref(pf), C.byref(pf), C.byref(prev), C.byref(ts),
                                       C.byref(hb), C.byref(ics), C.byref(box))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            if st != 0:
                raise RuntimeError(lib.c21cm_last_error().decode())
            best = dt if best is None else min(best, dt)
        return best

    dev = (density, n_ion, torch.ones_like(density), torch.zeros_like(density), torch.zeros_like(density),
           torch.zeros_like(density))
    run(dev, 1)
    ms_dev = run(dev, 3)
    xh_dev = float(dev[2].mean())
    host = tuple(np.ascontiguousarray(t.cpu().numpy()) for t in dev)
    host[2][...] = 1.0
    host[3][...] = 0.0
    ms_host = run(host, 2)
    return {"what": "ComputeIonizedBox(z=9, L-INTEGRAL grids) through the drop-in entry point, best of 3 / 2 calls",
            "device_arrays_ms": ms_dev, "host_arrays_ms": ms_host, "h2d_d2h_ms": ms_host - ms_dev,
            "bytes_over_pcie": 6 * 4 * float(n) ** 3, "global_xH": xh_dev,
            "global_xH_host_arrays": float(host[2].mean())}


PASS_KERNELS = {
    0: "line_pass_kernel<{n},+1,3>  (pass X: x-lines of both grids x streamed W(kR) tables)",
    1: "line_pass_kernel<{n},+1,0>  (pass Y: y-lines of both grids)",
    2: "zw_ionise_kernel<16,false,16> (wave-level pass Z of both grids + f_coll sum + barrier)",
    4: "window_table_kernel         (W(kR) of one radius for both windows, evaluated in fp64, stored as float)",
    6: "line_pass_kernel<{n},+1,5>  (pass X of TWO radii: each tile of both grids read once, windowed and transformed twice)",
    7: "line_pass_kernel<{n},+1,6>  (pass X: x-lines of both grids, W(kR) evaluated in the kernel from node tables in LDS)",
    8: "line_pass_kernel<{n},+1,7>  (pass X of TWO radii, W(kR) evaluated in the kernel: no window tables)",
}
# the key of each kernel in profiles/pmc_rNN.json
PMC_KEYS = {0: "pass_x_window", 1: "pass_y", 2: "pass_z_fused", 4: "window_tables", 6: "pass_x_pair",
            7: "pass_x_eval", 8: "pass_x_pair_eval"}


def kernel_roofline(args, spec, torch):
    """Live HIP-event timing of each hand-written pass kernel of the R loop, on torch's current
    stream (the stream every kernel of the step is launched on), with its ALGORITHMIC bytes:
      pass X / pass Y  read + write of both split k-space grids       2 * 2 * S
      pass X, 2 radii  one read, two writes of both grids             2 * 3 * S
      fused pass Z     read of both grids + uint8 mask read + write   2 * S + 2 * N
    S = 8 * (N/2 + nx*ny) bytes (split layout), N = cells.  The window tables pass X also
    reads (2 x 4 (n/2+1)^3 bytes, float entries) are overhead, not algorithmic bytes."""
    import ctypes as C

    n = args.hii_dim
    lib = importlib.import_module("21cmfast_amd").load()
    lib.c21hip_bench_pass.restype = C.c_int
    lib.c21hip_bench_pass.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                      C.c_double, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    N = float(n) ** 3
    S = 8.0 * (N / 2 + n * n)
    alg = {0: 4 * S, 1: 4 * S, 2: 2 * S + 2 * N, 4: 0.0, 6: 6 * S, 7: 4 * S, 8: 6 * S}
    # windows evaluated inside pass X (round 3; C21CM_WINDOWS=table keeps the streamed tables):
    # the step then runs kernels 7 / 8 instead of 0 / 6 and launches no table kernel
    lib.c21hip_wev_applicable.restype = C.c_int
    evaluated = bool(lib.c21hip_wev_applicable(int(spec.hii_filter), int(spec.stars_filter), 2, n, n, n))
    lib.c21hip_pair_sweep_supported.restype = C.c_int
    paired = (os.environ.get("C21CM_PAIR_RADII", "1") != "0" and bool(lib.c21hip_pair_sweep_supported(n))
              and spec.fcoll_mode == importlib.import_module("21cmfast_amd.workloads").FCOLL_STARS)
    n_fused = spec.n_radii - 1  # launches per step (radius index 0 is the final sweep)
    launches = {0: n_fused % 2 if paired else n_fused, 6: n_fused // 2 if paired else 0,
                1: n_fused, 2: n_fused, 4: n_fused}
    launches[7], launches[8] = launches[0], launches[6]
    R_mid = spec.R[spec.n_radii // 2]
    out = {}
    kinds = ((7, 1, 2) + ((8,) if paired else ())) if evaluated else ((0, 1, 2, 4) + ((6,) if paired else ()))
    for kind in kinds:
        ms = C.c_float()
        st = lib.c21hip_bench_pass(kind, n, int(spec.hii_filter), int(spec.stars_filter), R_mid,
                                   float(spec.mfp_meandens) or 1.0, spec.box_len, 20, stream,
                                   C.byref(ms))
        if st != 0:
            return None
        out[kind] = {"kernel": PASS_KERNELS[kind].replace("{n}", str(n)), "ms": ms.value, "alg_bytes": alg[kind],
                     "GBs": alg[kind] / ms.value / 1e6, "launches_per_step": launches[kind],
                     "ms_per_step": ms.value * launches[kind]}
    return out


def pmc_traffic(n=512):
    """HBM bytes per launch of the pass kernels from the newest committed rocprofv3 PMC passes
    (profiles/pmc_rNN.json, produced by tools/collect_pmc.py on the GPU box); None if absent.
    The profile records a hash of the kernel sources it was collected from; `stale` tells
    whether the kernels running now are those."""
    files = sorted((ROOT / "profiles").glob("pmc_r*.json" if n == 512 else f"pmc{n}_r*.json"))
    if not files:
        return None
    try:
        pmc = json.loads(files[-1].read_text())
    except (OSError, ValueError):
        return None
    pmc["file"] = f"profiles/{files[-1].name}"
    try:
        sys.path.insert(0, str(ROOT / "tools"))
        from collect_pmc import kernel_sources_sha

        pmc["stale"] = pmc.get("kernel_sources_sha16") != kernel_sources_sha()
    except Exception:
        pmc["stale"] = None
    return pmc


def run_icpf(args, torch, pkg):
    """BASELINE config 2: InitialConditions + PerturbedField (HII_DIM = 256, DIM = 512 by default; device
    arrays, Philox modes -- the reference's own random stream is a host-serial draw), one step = both
    calls.  Roofline on SURVEY 8(d)'s figures: the IC pipeline is 15 hi-res FFTs + 12 k-sweeps + 7 filter
    sweeps + 9 gathers = 68 passes over a DIM^3 spectrum (36.7 GB at DIM = 512), the deposit nominally
    N_h (4 + 8 x 16) B (one read and eight fp64 read-modify-writes per particle)."""
    sys.path.insert(0, str(ROOT / "tests"))
    api = importlib.import_module("21cmfast_amd.grid_api")
    from test_oracle_ics import ics_spec
    from test_oracle_perturb import perturb_spec

    hii = args.hii_dim // 2 if args.hii_dim == 512 else args.hii_dim
    dim = 2 * hii
    L = 1.5 * hii
    spec = ics_spec(dim, hii, box_len=L, seed=12345)
    pspec = perturb_spec(2, dim=dim, dim_z=dim, hii_dim=hii, hii_dim_z=hii, box_len=L, box_len_z=L,
                         growth_factor=0.127, init_growth_factor=0.0042, dDdt_over_D=2e-17)
    state = {"ics": api.ics_grids(spec, device="cuda")}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_ic, t_pf = [], []

    def step(record):
        ev[0].record()
        state["ics"] = api.ics_grids(spec, state["ics"], device="cuda")
        ev[1].record()
        state["out"] = api.perturb_grids(pspec, state["ics"])
        ev[2].record()
        if record:
            torch.cuda.synchronize()
            t_ic.append(ev[0].elapsed_time(ev[1]))
            t_pf.append(ev[1].elapsed_time(ev[2]))

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    ic_ms, pf_ms = sum(t_ic) / len(t_ic), sum(t_pf) / len(t_pf)
    nk = dim * dim * (dim // 2 + 1)
    ic_bytes = 68 * 8.0 * nk
    cic_bytes = float(dim) ** 3 * (4 + 8 * 16)
    out = {
        "metric": f"InitialConditions + PerturbField hi-res cells/sec (HII_DIM={hii}, DIM={dim})",
        "value": float(dim) ** 3 / (ms * 1e-3), "unit": "cells/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE config 2: InitialConditions + PerturbedField, HII_DIM={hii}, DIM={dim}, "
                               "2LPT, device-resident arrays, counter-based (Philox) modes",
                   "hii_dim": hii, "dim": dim, "ics_ms": ic_ms, "perturb_ms": pf_ms,
                   "random_stream": "Philox-4x32-10 counters on the device; the reference-compatible GSL streams "
                                    "(same seed -> upstream's universe) are drawn serially on the host as upstream "
                                    "draws them and are NOT what this line times: 1.5 s at DIM = 1024 "
                                    "(profiles/r06_config5_timing.json: ics_ms)",
                   "density_std": float(state["out"]["density"].std())},
        "roofline": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0, "achievable_GBs": args.achievable_gbs,
                     "kernel": "the InitialConditions pipeline as a whole (native line passes, k-space "
                               "operators in pass X; 68 passes of SURVEY 8(d))",
                     "alg_bytes_per_launch": ic_bytes, "ms_per_launch": ic_ms,
                     "achieved": ic_bytes / (ic_ms * 1e-3) / 1e9, "frac": ic_bytes / (ic_ms * 1e-3) / 8e12,
                     "traffic": None,
                     "perturb_field": {"nominal_cic_bytes": cic_bytes, "ms": pf_ms,
                                       "note": "SURVEY 8(d)'s nominal N_h (4 + 8 x 16) B: the per-velocity-cell "
                                               "deposit keeps its read-modify-writes in LDS (PMC summary: "
                                               "profiles/r04_pmc_cic.json)",
                                       "nominal_GBs": cic_bytes / (pf_ms * 1e-3) / 1e9}},
    }
    # HBM bytes of one IC + PerturbedField step from the PMC passes of this command (tools/job_profiles.sh ->
    # profiles/pmc_icpf_rNN.json; None when no such profile is in the tree)
    pmc_files = sorted((ROOT / "profiles").glob("pmc_icpf_r*.json"))
    if pmc_files:
        try:
            pmc = json.loads(pmc_files[-1].read_text())
            out["roofline"]["traffic"] = pmc["hbm_bytes_per_step"]
            out["roofline"]["traffic_scope"] = ("one step = InitialConditions + PerturbedField (all kernels), "
                                                f"profiles/{pmc_files[-1].name}: {pmc['source']}")
            out["roofline"]["actual_hbm_frac"] = pmc["hbm_bytes_per_step"] / (ms * 1e-3) / 8e12
        except (OSError, ValueError, KeyError):
            pass
    if not args.no_cpu_baseline:
        oracle = importlib.import_module("oracle.oracle")
        cores = min(os.cpu_count() or 1, 64)
        oracle.set_threads(cores)
        chii = hii if cores >= 16 else hii // 2
        cspec = ics_spec(2 * chii, chii, box_len=1.5 * chii, seed=12345)
        cps = perturb_spec(2, dim=2 * chii, dim_z=2 * chii, hii_dim=chii, hii_dim_z=chii, box_len=1.5 * chii,
                           box_len_z=1.5 * chii, growth_factor=0.127, init_growth_factor=0.0042,
                           dDdt_over_D=2e-17)
        t0 = time.perf_counter()
        ref = oracle.ics_grids(cspec)
        oracle.perturb_grids(cps, ref)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": float(2 * chii) ** 3 / dt, "unit": "cells/s", "cores": cores,
                               "kind": "port", "seconds": dt,
                               "sample": f"the CPU oracle's InitialConditions + PerturbedField at HII_DIM={chii}, "
                                         f"DIM={2 * chii}, {cores} threads"}
        out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def main():
    args = parse_args()
    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a
    # version banner through C stdio, flushed at exit, i.e. after anything Python printed), so
    # file descriptor 1 is pointed at stderr for the whole run and the JSON line goes to the
    # saved descriptor at the very end.
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    local_rank %= max(1, torch.cuda.device_count())  # ranks may share a GPU in the gloo test
    torch.cuda.set_device(local_rank)
    dist = None
    sharded = world > 1 or args.force_shard
    if sharded:
        import torch.distributed as dist

        if args.force_shard and "RANK" not in os.environ:
            import socket

            with socket.socket() as sock:  # any free port: this is a one-rank group
                sock.bind(("127.0.0.1", 0))
                port = sock.getsockname()[1]
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(port))
            dist.init_process_group(args.backend, rank=0, world_size=1,
                                    device_id=torch.device("cuda", local_rank))
        elif args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    pkg = importlib.import_module("21cmfast_amd")
    pkg.load(require_gpu=True)
    if args.mode == "icpf":  # config 2: a single-GPU line of its own
        out = run_icpf(args, torch, pkg)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
        os.close(real_stdout)
        return
    W = importlib.import_module("21cmfast_amd.workloads")
    api = importlib.import_module("21cmfast_amd.grid_api")
    D = importlib.import_module("21cmfast_amd.distributed")

    n = args.hii_dim
    mode = W.FCOLL_STARS if args.mode == "stars" else W.FCOLL_ERFC
    G = 2 if mode == W.FCOLL_STARS else 1

    class Workload:
        """One box of the benchmark: spec, device-resident inputs, output buffers, and the step
        in its single-GPU and its sharded form."""

        def __init__(self, dim):
            self.n = dim
            self.spec = W.ionize_spec(dim, mode=mode, r_bubble_max=args.r_bubble_max)
            self.density = W.density_field_torch(dim, seed=12345)
            self.n_ion = W.nion_from_density(self.density) if mode == W.FCOLL_STARS else None
            self.buffers = api.IonizeBuffers(self.density, need_nion=mode != W.FCOLL_STARS)
            self.owner = D.owner_rank(self.spec.n_radii, world)
            self.first_cross = None
            self.report = {}

        def step_single(self):
            self.buffers.reset()
            _, _, rep = api.ionize_grids(self.spec, self.density, self.n_ion, buffers=self.buffers)
            self.report["rep"] = rep

        def step_sharded(self, shard_c, gather=False):
            # finish phase by cell slabs (round 5): every rank finishes the cells of its slab and holds
            # the complete scalars; the outputs stay slab-resident unless `gather` (all-gather of the
            # output slabs: whole boxes on every rank)
            self.slab = api.shard_slab_supported(self.spec) and os.environ.get("C21CM_SHARD_FINISH", "")[:1] != "o"
            if self.slab or rank == self.owner:
                self.buffers.reset()
            if shard_c:
                rep = D.sharded_ionize_c(self.spec, self.density, self.n_ion, self.buffers, rank, world,
                                         broadcast=gather)
            else:
                if self.first_cross is None:
                    self.first_cross = torch.zeros((self.n,) * 3, dtype=torch.uint8, device="cuda")
                if self.slab:
                    rep = D.sharded_ionize_slabs(self.spec, self.density, self.n_ion, self.buffers,
                                                 self.first_cross, rank, world, gather_outputs=gather)
                    if rank != self.owner:
                        rep = None
                else:
                    rep = D.sharded_ionize(self.spec, self.density, self.n_ion, self.buffers,
                                           self.first_cross, rank, world)
            if rep is not None:
                self.report["rep"] = rep

    def fence():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(stepfn, steps, warmup, collective=True):
        """W untimed + K timed steps between fences; ms per step, the MAX over the ranks when
        `collective` (every rank runs the steps), this rank's own clock otherwise."""
        for _ in range(warmup):
            stepfn()
        fence() if collective else torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            stepfn()
        fence() if collective else torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if sharded and collective:
            t = torch.tensor([el], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        return el / steps * 1e3

    def shard_phase_report():
        """shard phase / exchange / finish of the last sharded step on every rank (C-level
        exchange), gathered to all; and the rank count RCCL itself reports."""
        import ctypes as C

        lib = pkg.load()
        lib.c21cm_shard_last_phases.restype = C.c_int
        lib.c21cm_shard_comm_count.restype = C.c_int
        ph = (C.c_double * 3)()
        ok = lib.c21cm_shard_last_phases(ph) == 0
        mine = torch.tensor([ph[0], ph[1], ph[2]] if ok else [float("nan")] * 3, device="cuda",
                            dtype=torch.float64)
        allp = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        return ([[round(float(v), 3) for v in t.tolist()] for t in allp],
                int(lib.c21cm_shard_comm_count()))

    def single_gpu_same_run(wl, steps):
        """The unsharded pass on rank 0 of THIS run (same box, same binaries, same clocks), the
        other ranks idle between two barriers: the denominator of `speedup`."""
        ms = None
        fence()
        if rank == 0:
            ms = timed(wl.step_single, steps, 1, collective=False)
        fence()
        t = torch.tensor([ms if ms is not None else 0.0], device="cuda", dtype=torch.float64)
        dist.broadcast(t, src=0)
        return t.item()

    wl = Workload(n)
    spec, density, n_ion, buffers, owner = wl.spec, wl.density, wl.n_ion, wl.buffers, wl.owner
    # the C-level exchange needs a communicator the library can form: RCCL (nccl backend, one GPU per rank),
    # or -- several ranks on ONE GPU, where RCCL refuses -- the test-only stand-in named by C21CM_RCCL_LIB
    # (tests/shim/rccl_shim.c: shared memory; executes the same sends and receives, times nothing real)
    shard_c = sharded and args.shard_impl == "c" and (args.backend == "nccl" or bool(os.environ.get("C21CM_RCCL_LIB")))
    shard_note = None
    if shard_c:
        # the library's own communicator, bootstrapped once through torch.distributed.  One probe
        # call decides COLLECTIVELY whether it works here; otherwise every rank takes the
        # torch.distributed exchange (same kernels, the reduce issued from Python).
        ok, why = 1, ""
        try:
            api.shard_init_from_torch()
            buffers.reset()
            D.sharded_ionize_c(spec, density, n_ion, buffers, rank, world)
            torch.cuda.synchronize()
        except Exception as exc:  # noqa: BLE001 -- any failure means "use the other exchange"
            ok, why = 0, f"{type(exc).__name__}: {exc}"
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            shard_c = False
            shard_note = "torch (C-level RCCL exchange unavailable" + (f": {why}" if why else " on another rank") + ")"
            print(f"[bench rank {rank}] falling back to the torch.distributed exchange. {why}",
                  file=sys.stderr, flush=True)
            try:
                api.shard_finalize()
            except Exception:  # noqa: BLE001
                pass
    last_report = wl.report

    def step():
        if not sharded:
            wl.step_single()
        else:
            # whole boxes on every rank: what ComputeIonizedBox leaves by default on a communicator (round 6,
            # ADVICE r5); the slab-resident opt-in is timed beside it
            wl.step_sharded(shard_c, gather=True)

    # the first call of a size pays for the workspace allocations and the node tables: reported, not timed
    t_cold = time.perf_counter()
    step()
    torch.cuda.synchronize()
    first_call_ms = (time.perf_counter() - t_cold) * 1e3
    # Where the work spectra sit in HBM (csrc/host/placement.c).  The library allocates plainly unless asked (a walk
    # costs 0.01-5 s once per process and box size and buys 1-4 ms per 512^3 call: the caller's decision).  This
    # benchmark measures steady-state throughput, so it asks -- after timing the library's default as well:
    # `default_allocation` = plain hipMalloc, the headline = with the second spectrum of each two-grid sweep placed
    # by timed launches (C21CM_WS_PLACE=0 in the environment keeps the plain allocations for the headline too).
    default_alloc = None
    if os.environ.get("C21CM_WS_PLACE", "") == "" and not sharded:
        ms_plain = timed(step, max(3, args.steps // 2), 1)
        default_alloc = {"ms_per_step": ms_plain, "value": float(n) ** 3 / (ms_plain * 1e-3), "unit": "cells/s",
                         "what": "the library's default: plain allocations, no placement walk"}
        os.environ.setdefault("C21CM_WS_PLACE_MS", "20000")  # (a throughput run waits for its walk; default 300 ms)
        api.placement_set(1)
        t_walk = time.perf_counter()
        step()
        torch.cuda.synchronize()
        default_alloc["placement_call_ms"] = (time.perf_counter() - t_walk) * 1e3
    elif os.environ.get("C21CM_WS_PLACE", "") == "":
        # sharded: every rank opts in for its own device (a walk only happens where the rank has the GPU to itself)
        os.environ.setdefault("C21CM_WS_PLACE_MS", "20000")
        api.placement_set(1)
        step()
        torch.cuda.synchronize()
    placement = api.placement_report()
    ms_per_step = timed(step, args.steps, args.warmup)
    ms_slab_resident = None
    if sharded and getattr(wl, "slab", False):
        ms_slab_resident = timed(lambda: wl.step_sharded(shard_c, gather=False), args.steps, 1)
        step()  # (the report / buffers of the headline form)
    cells = float(n) ** 3
    value = cells / (ms_per_step * 1e-3)

    # per-rank phase times of the last sharded step (C-level exchange): shard phase / exchange /
    # finish on every rank, gathered to rank 0, so that a SCALE line can be read
    shard_phases, rccl_ranks = None, None
    if sharded and shard_c:
        shard_phases, rccl_ranks = shard_phase_report()

    # the finishing rank holds the result: hand its global x_HI to rank 0 for the JSON line
    global_xh = None
    if sharded:
        rep_o = last_report.get("rep")
        gx = torch.tensor([rep_o.global_xH if rep_o is not None else 0.0], device="cuda",
                          dtype=torch.float64)
        dist.broadcast(gx, src=owner)
        global_xh = gx.item()
    elif last_report.get("rep") is not None:
        global_xh = last_report["rep"].global_xH

    # ---- sharded runs: the single-GPU time of the SAME run, and BASELINE config 4
    same_run, config4 = None, None
    if sharded:
        saved_rep = dict(last_report)
        ms_single = single_gpu_same_run(wl, max(2, min(args.steps, 5)))
        last_report.clear()
        last_report.update(saved_rep)
        same_run = {"ms_per_step": ms_single, "n_gpus": 1,
                    "what": "the unsharded pass on rank 0 inside this run (other ranks idle)",
                    "speedup": ms_single / ms_per_step}
        n4 = args.config4_dim
        if n4 and n4 != n:
            wl.first_cross = None
            wl4 = Workload(n4)
            k4 = max(1, args.config4_steps)
            ms4 = timed(lambda: wl4.step_sharded(shard_c, gather=True), k4, 1)
            ph4, cnt4 = shard_phase_report() if shard_c else (None, None)
            # the same with slab-resident outputs (no all-gather of the three output slabs: the opt-in)
            ms4_slab = timed(lambda: wl4.step_sharded(shard_c, gather=False), k4, 1) if wl4.slab else None
            wl4.step_sharded(shard_c, gather=True)
            gx4 = torch.tensor([wl4.report["rep"].global_xH if wl4.report.get("rep") is not None else 0.0],
                               device="cuda", dtype=torch.float64)
            dist.broadcast(gx4, src=wl4.owner)
            ms4_single = single_gpu_same_run(wl4, max(1, min(k4, 3)))
            gx4_single = wl4.report["rep"].global_xH if rank == 0 and wl4.report.get("rep") is not None else None
            config4 = {
                "workload": f"ComputeIonizedBox single-z, HII_DIM={n4}, {wl4.spec.n_radii} filter steps, "
                            f"G={G}, R loop sharded x{world}",
                "hii_dim": n4, "n_radii": wl4.spec.n_radii, "n_gpus": world, "steps": k4, "warmup": 1,
                "ms_per_step": ms4, "value": float(n4) ** 3 / (ms4 * 1e-3), "unit": "cells/s",
                "single_gpu_same_run": {"ms_per_step": ms4_single, "steps": max(1, min(k4, 3))},
                "speedup": ms4_single / ms4,
                "finish": "by cell slabs, whole boxes all-gathered to every rank" if wl4.slab
                          else "on the owner rank, its box broadcast",
                **({"ms_per_step_slab_resident": ms4_slab,
                    "speedup_slab_resident": ms4_single / ms4_slab} if ms4_slab else {}),
                "global_xH": gx4.item(), "global_xH_single_gpu": gx4_single,
                **({"shard_phases_ms_per_rank": ph4, "rccl_comm_count": cnt4} if ph4 is not None else {}),
            }
            del wl4
            torch.cuda.empty_cache()

    out = None
    if rank == 0:
        native = bool(pkg.load().c21hip_fft_is_native(n, n, n))
        # Lagrangian grids on the native transform: radius index 0 applies no window, so its
        # filtered emissivity is the input itself and the step is ONE sweep (mask + barrier +
        # partial ionisation + post-loop: 17N read, 12N written) instead of a transform round
        # trip; only the bytes actually owed are counted (C21CM_R0_ROUNDTRIP=1 restores it).
        r0_direct = native and G == 2 and os.environ.get("C21CM_R0_ROUNDTRIP", "0") != "1"
        alg_loop = algorithmic_bytes_per_radius(cells, G) * (spec.n_radii - (1 if r0_direct else 0))
        if r0_direct:
            alg_loop += 29.0 * cells
        # `achievable_GBs`: what a plain float4 copy kernel moves on these boxes (tools/copy_bench.hip,
        # profiles/r05_copy_bench.txt: 6.2 TB/s read + write; 5.4-5.8 with 4-8 float4 per thread,
        # 4.7 through hipMemcpy) -- the ceiling a streaming kernel can be held against; `frac` stays
        # against the 8 TB/s peak as the contract says
        roof = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achievable_GBs": args.achievable_gbs,
                "traffic": None}
        kern = None if (world > 1 or args.no_kernel_roofline or not native) else \
            kernel_roofline(args, spec, torch)
        if kern:
            # The launch durations INSIDE the step: one more step with HIP events around every pass
            # launch on its stream (c21hip_ktime_*).  A kernel between its neighbours of the R loop
            # runs slower than the same kernel launched back to back (pass Y: 0.49 against
            # 0.41-0.47 ms), and it is the in-loop average that the rocprofv3 kernel stats show, so
            # that is what `achieved` is computed from; the back-to-back figure stays as
            # `ms_isolated`.
            import ctypes as C

            lib = pkg.load()
            lib.c21hip_ktime_report.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
            saved_report = dict(last_report)  # the R-loop time reported below is the timed steps'
            lib.c21hip_ktime_enable(1)
            step()
            torch.cuda.synchronize()
            last_report.clear()
            last_report.update(saved_report)
            for kind, k in kern.items():
                tot, cnt = C.c_double(), C.c_int()
                if lib.c21hip_ktime_report(kind, C.byref(tot), C.byref(cnt)) == 0 and cnt.value > 0:
                    k["ms_isolated"] = k["ms"]
                    k["ms"] = tot.value / cnt.value
                    k["launches_per_step"] = cnt.value
                    k["GBs"] = k["alg_bytes"] / k["ms"] / 1e6
                    k["ms_per_step"] = tot.value
                    k["timing"] = "HIP events around each launch inside one step of the R loop"
            if G == 1:
                # one filtered grid (--mode erfc): the line passes of this loop move ONE split spectrum
                # (read + write: 2 S), and the two-grid fused pass Z is not part of it -- its pass Z is
                # zw_c2r_kernel<16, 7, 16> (closed form + banded barrier), timed as kind 12 below
                S1 = 8.0 * (cells / 2 + n * n)
                for kind in list(kern):
                    k = kern[kind]
                    if "timing" not in k:
                        del kern[kind]
                        continue
                    k["alg_bytes"] = 2 * S1
                    k["GBs"] = k["alg_bytes"] / k["ms"] / 1e6
                    k["kernel"] = k["kernel"].replace("of both grids", "of the one grid")
                tot, cnt = C.c_double(), C.c_int()
                if lib.c21hip_ktime_report(12, C.byref(tot), C.byref(cnt)) == 0 and cnt.value > 0:
                    # pass Z of the one grid + closed-form f_coll, its barrier decided in the sweep (banded):
                    # the spectrum read + the mask row read (the dense-sweep launches of the first radii,
                    # which also write the f_coll grid, are in the average)
                    zb = S1 + cells
                    kern[12] = {"kernel": "zw_c2r_kernel<16,7,16> (pass Z of the one grid + closed-form f_coll + "
                                          "banded barrier; a few launches per step are the dense-sweep EPI 2)",
                                "ms": tot.value / cnt.value, "alg_bytes": zb,
                                "GBs": zb / (tot.value / cnt.value) / 1e6, "launches_per_step": cnt.value,
                                "ms_per_step": tot.value,
                                "timing": "HIP events around each launch inside one step of the R loop",
                                "bound_note": "fp64-issue bound (erfc), not byte bound: see DESIGN section 4"}
            lib.c21hip_ktime_enable(0)
            # dominant kernel = the one with the largest share of the R loop (launch time x
            # launches per step; agrees with the kernel-trace stats in profiles/)
            dom_kind = max((k for k in kern if k != 4), key=lambda k: kern[k]["ms_per_step"])
            dom = kern[dom_kind]
            roof.update({"kernel": dom["kernel"], "achieved": dom["GBs"],
                         "frac": dom["GBs"] / HBM_PEAK_GBS, "ms_per_launch": dom["ms"],
                         "alg_bytes_per_launch": dom["alg_bytes"],
                         "launches_per_step": dom["launches_per_step"],
                         "other_kernels": [kern[k] for k in sorted(kern) if k != dom_kind]})
            pmc = pmc_traffic(n) if G == 2 else None  # (the PMC passes profile the two-grid launches)
            if pmc:
                per = pmc.get("kernels", {}).get(PMC_KEYS.get(dom_kind))
                roof["traffic"] = per["hbm_bytes"] if per else None
                roof["traffic_source"] = f"{pmc.get('file')}: {pmc.get('source')}"
                # False: collected from exactly the kernel sources running now
                roof["traffic_profile_stale"] = pmc.get("stale")
        rep = last_report.get("rep")
        # whole R loop against the SURVEY 8(d) contract: (20G + 8) * N bytes per radius
        loop = {"alg_bytes": alg_loop,
                "definition": "R loop of one step, (20G+8)*N algorithmic bytes per radius"
                              + (" for indices > 0, 29N for the index-0 sweep" if r0_direct else "")}
        if not sharded and rep is not None and rep.ms_rloop > 0:
            loop.update({"ms": rep.ms_rloop, "ms_preloop": rep.ms_preloop,
                         "ms_postloop": rep.ms_postloop,
                         "GBs": alg_loop / (rep.ms_rloop * 1e-3) / 1e9})
        else:
            loop.update({"ms": ms_per_step, "GBs": alg_loop / (ms_per_step * 1e-3) / 1e9})
        loop["frac"] = loop["GBs"] / HBM_PEAK_GBS
        # ... and what the loop actually moves (VERDICT r5 weak point 4): the PMC bytes per launch of every pass
        # kernel x its launches in the step / the loop's time.  `frac` prices the 8(d) CONTRACT bytes (the design
        # moves fewer: two radii per pass-X read, no separate threshold sweep), `actual_hbm_frac` is the HBM
        # utilisation proper.
        if kern and G == 2:
            pmc_all = pmc_traffic(n)
            moved, missing = 0.0, []
            for kind, k in kern.items():
                per = (pmc_all or {}).get("kernels", {}).get(PMC_KEYS.get(kind))
                if per:
                    moved += per["hbm_bytes"] * k["launches_per_step"]
                elif k.get("alg_bytes"):
                    missing.append(PMC_KEYS.get(kind))
            if pmc_all and moved and not missing:
                loop["actual_hbm_bytes"] = moved
                loop["actual_hbm_GBs"] = moved / (loop["ms"] * 1e-3) / 1e9
                loop["actual_hbm_frac"] = loop["actual_hbm_GBs"] / HBM_PEAK_GBS
                loop["actual_hbm_source"] = (f"{pmc_all.get('file')} (per-launch FETCH_SIZE x 2 + WRITE_SIZE of the "
                                             "pass kernels) x launches per step; the index-0 sweep and the pre/post "
                                             "loops are not in it")
        roof["r_loop"] = loop
        if "achieved" not in roof:
            roof.update({"kernel": "R loop (all kernels)", "achieved": loop["GBs"],
                         "frac": loop["frac"]})
        out = {
            "metric": f"IonizeBox cells/sec ({n}^3 HII_DIM, {spec.n_radii} filter radii)",
            "value": value, "unit": "cells/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"ComputeIonizedBox single-z, HII_DIM={n}, {spec.n_radii} filter "
                            f"steps, G={G} ({'delta tophat + n_ion exp-MFP' if G == 2 else 'delta sharp-k, erfc f_coll'}), "
                            "first snapshot, device-resident inputs",
                "hii_dim": n, "n_radii": spec.n_radii, "filtered_grids": G,
                "parallelism": "single GPU" if not sharded else
                f"R-loop sharded x{world} + "
                + (("packed first crossings by slab + slab finish on every rank, over RCCL inside the C "
                    "library (c21cm_ionize_sharded)" if getattr(wl, "slab", False) else
                    "1-bit mask gather over RCCL inside the C library (c21cm_ionize_sharded)") if shard_c
                   else (f"slab exchange + slab finish via torch.distributed ({args.backend})"
                         if getattr(wl, "slab", False) else
                         f"{'RCCL' if args.backend == 'nccl' else 'gloo'} uint8 max-reduce via torch.distributed")),
                **({"shard_impl_note": shard_note} if shard_note else {}),
                **({"shard_outputs": "whole boxes on every rank (the ABI default)",
                    "ms_per_step_slab_resident": ms_slab_resident} if sharded else {}),
                **({"shard_transport": "tests/shim/librccl_shim.so (shared memory, ranks share one GPU): "
                                       "the exchange code is executed, xGMI is not measured"}
                   if shard_c and os.environ.get("C21CM_RCCL_LIB") else {}),
                **({"shard_phases_ms_per_rank": shard_phases, "shard_phases": "shard phase, exchange "
                    "(incl. waiting for the slowest peer), finish -- device time of the last step",
                    "rccl_comm_count": rccl_ranks} if shard_phases is not None else {}),
                "fft": "native" if native else "rocfft",
                # where the work spectra sit in HBM (csrc/host/placement.c; decided once, in the warm-up)
                "work_spectra_placement": "plain hipMalloc (the library's default)"
                if (placement or {}).get("outcome", "").startswith(("off", "other", "another")) or not placement
                else "second spectrum of each two-grid sweep chosen by timed launches (opt-in: c21cm_placement_set(1)); "
                     "the library's default is timed in default_allocation",
                "global_xH": global_xh,
            },
            "roofline": roof,
            "first_call_ms": first_call_ms,
            "placement": placement,
            **({"default_allocation": default_alloc} if default_alloc else {}),
        }
        if same_run is not None:
            out["single_gpu_same_run"] = same_run
            out["speedup"] = same_run["speedup"]
        if config4 is not None:
            out["config4"] = config4
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_abi and mode == W.FCOLL_STARS:
        try:
            out["abi"] = abi_timing(n, args, torch, pkg, W)
            out["abi_ms"] = out["abi"]["device_arrays_ms"]
            out["h2d_d2h_ms"] = out["abi"]["h2d_d2h_ms"]
        except Exception as exc:  # noqa: BLE001 -- a diagnostic, never the reason a bench line is lost
            out["abi"] = {"error": f"{type(exc).__name__}: {exc}"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        def gpu_run(spec_c):
            """the HIP pass on the oracle sample's spec: inputs as host copies + outputs"""
            nn = spec_c.hii_dim
            dens = density if nn == n else W.density_field_torch(nn, seed=12345)
            nion = (n_ion if nn == n else W.nion_from_density(dens)) if mode == W.FCOLL_STARS else None
            b, _, r = api.ionize_grids(spec_c, dens, nion)
            torch.cuda.synchronize()
            return {"density": dens.cpu().numpy(), "n_ion": None if nion is None else nion.cpu().numpy(),
                    "neutral_fraction": b.neutral_fraction.cpu().numpy(), "z_reion": b.z_reion.cpu().numpy(),
                    "report": r}

        out["cpu_baseline"] = cpu_baseline(args, W, gpu_run)
        out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        if "parity" in out["cpu_baseline"]:
            par = out["cpu_baseline"].pop("parity")
            out[f"parity_{par['box']}"] = par
    if shard_c:
        api.shard_finalize()
    if sharded:
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.close(real_stdout)


if __name__ == "__main__":
    main()
