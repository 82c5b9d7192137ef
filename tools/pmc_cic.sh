#!/bin/bash
# Hardware counters of the PerturbedField deposit kernels at DIM = 1024 -> HII_DIM = 512
# (tools/time_cic.py: cell / tiled / direct on the same synthetic fields), written to
# profiles/r04_pmc_cic.json.  Separate rocprofv3 --pmc passes (SQ counters 8 per pass, FETCH_SIZE and
# WRITE_SIZE each alone: MI355X_MICROARCH.md, HBM section), every pass under its own timeout.
# usage (GPU box): tools/pmc_cic.sh [out.json]
REPO=$PWD
OUT=${1:-$REPO/gpurun_out/r04_pmc_cic.json}
export TMPDIR=/tmp
cd /tmp
rm -rf $REPO/gpurun_out/pmc_cic4
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_cic4/p$i -o cic -- \
      env PYTHONPATH=$REPO python $REPO/tools/time_cic.py 512 1024 > /dev/null 2>&1
done
python - "$REPO" "$OUT" <<'PY'
import csv, glob, json, sys, collections
repo, out = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(f"{repo}/gpurun_out/pmc_cic4/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        key = ("cell" if "cic_cell_kernel" in k else "tiled" if "cic_scatter_tiled" in k else
               "direct" if "cic_scatter_kernel" in k else None)
        if key:
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
res = {"source": "rocprofv3 --pmc (4 separate passes) around tools/time_cic.py 512 1024; means per launch; "
                 "FETCH_SIZE / WRITE_SIZE in KiB, FETCH x2 for 16-byte-per-lane streaming reads only "
                 "(gfx950 correction) -- this kernel reads 4 bytes per lane, so both readings are given",
       "particles": 1024**3, "kernels": {}}
for key, counters in acc.items():
    m = {c: sum(v) / len(v) for c, v in counters.items()}
    d = {"launch_ms_under_pmc": sum(dur[key]) / len(dur[key]), "counters": m}
    if "SQ_WAVE_CYCLES" in m and m.get("SQ_WAVE_CYCLES"):
        wc = m["SQ_WAVE_CYCLES"]
        d["valu_active_frac_of_wave_cycles"] = m.get("SQ_ACTIVE_INST_VALU", 0) / wc
        d["lds_active_frac_of_wave_cycles"] = m.get("SQ_ACTIVE_INST_LDS", 0) / wc
        d["lds_issue_stall_frac"] = m.get("SQ_WAIT_INST_LDS", 0) / wc
        d["valu_insts_per_particle"] = m.get("SQ_INSTS_VALU", 0) * 64 / 1024**3
        d["lds_insts_per_particle"] = m.get("SQ_INSTS_LDS", 0) * 64 / 1024**3
    if "SQ_WAIT_ANY" in m:
        tot = m["SQ_WAIT_ANY"] + m.get("SQ_WAIT_INST_ANY", 0) + m.get("SQ_ACTIVE_INST_ANY", 0)
        if tot:
            d["wave_parked_frac"] = m["SQ_WAIT_ANY"] / tot
            d["issue_stall_frac"] = m.get("SQ_WAIT_INST_ANY", 0) / tot
            d["issuing_frac"] = m.get("SQ_ACTIVE_INST_ANY", 0) / tot
    if "FETCH_SIZE" in m:
        d["fetch_bytes_raw"] = m["FETCH_SIZE"] * 1024
        d["fetch_bytes_x2"] = m["FETCH_SIZE"] * 2048
    if "WRITE_SIZE" in m:
        d["write_bytes"] = m["WRITE_SIZE"] * 1024
    res["kernels"][key] = d
res["algorithmic_bytes"] = {"density_read": 4 * 1024**3, "velocity_read_6_grids": 6 * 4 * 512**3,
                            "output_double_grid": 8 * 512**3}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
PY
