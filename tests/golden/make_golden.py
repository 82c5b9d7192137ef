#!/usr/bin/env python
"""Regenerate the golden fixtures with the CPU oracle:  python tests/golden/make_golden.py

They pin the oracle against silent drift (tests/test_golden.py, CPU) and give the HIP path a
fixed target that does not depend on the oracle being rebuilt (tests/test_gpu_golden.py).
The oracle itself is pinned against the reference's known-answer tests, see DESIGN.md section 2.
"""

import importlib
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
sys.path.insert(0, str(HERE))

import cases  # noqa: E402

oracle = importlib.import_module("oracle.oracle")


def main():
    oracle.load()
    inp = cases.ionize_inputs()
    for kind in ("lagrangian", "erfc"):
        out = cases.ionize_outputs(lambda s, d, n, nn: oracle.ionize_grids(s, d, n, need_nion=nn), kind, inp)
        np.savez_compressed(HERE / f"ionize_{kind}_{cases.N_ION}.npz", **inp, **out)
    inp64 = cases.ionize_inputs(cases.N_ION_NATIVE)
    out64 = cases.ionize_outputs(lambda s, d, n, nn: oracle.ionize_grids(s, d, n, need_nion=nn),
                                 "lagrangian", inp64)
    np.savez_compressed(HERE / f"ionize_lagrangian_{cases.N_ION_NATIVE}.npz", **inp64, **out64)
    np.savez_compressed(HERE / "perturb_roll.npz", **cases.perturb_roll_outputs(oracle.perturb_grids))
    np.savez_compressed(HERE / "filters_delta.npz", **cases.filter_outputs(oracle.filter_grid))
    hd = cases.ics_input()
    out = cases.ics_perturb_outputs(oracle.new_ics_arrays, oracle.ics_grids, oracle.perturb_grids, hd)
    np.savez_compressed(HERE / "ics_perturb.npz", hires_density=hd, **out)
    ts_in = cases.tsfilter_inputs()
    ts_out = cases.tsfilter_outputs(oracle.fill_Rbox_grids, oracle.annular_filter_grids, ts_in)
    np.savez_compressed(HERE / "tsfilter.npz", **ts_in, **ts_out)
    for f in sorted(HERE.glob("*.npz")):
        print(f.name, f.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
