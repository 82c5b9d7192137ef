#!/bin/bash
# L2 plane hand-off probe (tools/l2_plane_probe.hip): timings, then HBM traffic per kernel from two
# separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB per launch).  GPU box only.
REPO=$PWD
BIN=$REPO/tools/bin/l2_plane_probe
OUT=$REPO/gpurun_out/l2_probe
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 60 $BIN 10 127 | tee $OUT/timing.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 90 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- $BIN 3 102 > /dev/null 2>&1
done
python3 - "$OUT" <<'PY' | tee $OUT/traffic.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0][-60:]][c].append(float(r["Counter_Value"]))
print("kernel, launches, FETCH_SIZE MB raw (x2 for 16-byte streaming reads), WRITE_SIZE MB   [algorithmic: 2147 MB read + 2147 MB hand-off written + 2147 MB hand-off read]")
for k, d in acc.items():
    f = d.get("FETCH_SIZE", [0]); w = d.get("WRITE_SIZE", [0])
    print(f"{k:62s} {len(f):3d}  fetch {sum(f)/len(f)/1024:9.1f}  (x2 = {sum(f)/len(f)/512:9.1f})   write {sum(w)/len(w)/1024:9.1f}")
PY
