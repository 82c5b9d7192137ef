#!/usr/bin/env python
"""Turn two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately as the
MI355X guide prescribes) into profiles/pmc_rNN.json.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_FETCH_SIZE -o pmc -- python bench.py ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_WRITE_SIZE -o pmc -- python bench.py ...
  python tools/collect_pmc.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE profiles/pmc_r01.json

Units and corrections (/opt/skills/guides/MI355X_MICROARCH.md, section HBM): the counters are
in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming
read (16 B per lane), so it is doubled; WRITE_SIZE is taken as is (uncalibrated).
"""

import csv
import json
import sys
from collections import defaultdict

KERNELS = {
    "pass_x_window": ("line_pass_kernel<512, 1, 1>", 1000),
    "pass_y": ("line_pass_kernel<512, 1, 0>", 1000),
    "pass_z_fused": ("z_c2r_ionise_kernel<512>", 1000),
}


def averages(directory, counter):
    acc = defaultdict(list)
    with open(f"{directory}/pmc_counter_collection.csv") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            for key, (needle, min_grid) in KERNELS.items():
                # main-block launches only (the Nyquist-plane launches use tiny grids)
                if needle in row["Kernel_Name"] and int(row["Grid_Size"]) >= 100000:
                    acc[key].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fetch = averages(fetch_dir, "FETCH_SIZE")
    write = averages(write_dir, "WRITE_SIZE")
    result = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), "
                        "KiB -> bytes, FETCH_SIZE x2 (gfx950 wide-read correction)",
              "kernels": {}}
    for key in KERNELS:
        if key in fetch and key in write:
            fb = fetch[key][0] * 1024 * 2
            wb = write[key][0] * 1024
            result["kernels"][key] = {"fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb,
                                      "launches": fetch[key][1]}
    dom = result["kernels"].get("pass_x_window")
    result["hbm_bytes_per_launch"] = dom["hbm_bytes"] if dom else None
    with open(out, "w") as f:
        json.dump(result, f, indent=1)
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
