#!/bin/bash
# SQ counters of the PerturbedField deposit kernel (diagnostic; every rocprofv3 under timeout)
REPO=$PWD
export TMPDIR=/tmp
cd /tmp
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  timeout 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_cic -o cic -- \
      env PYTHONPATH=$REPO python $REPO/tools/time_ic_pf.py 256 512 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
f = sorted(glob.glob("$REPO/gpurun_out/pmc_cic/**/cic_counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if "cic_scatter_tiled" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in acc.items(): print(k, v / max(n, 1))
PY
done
