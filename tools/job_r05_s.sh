#!/bin/bash
# the two placement modes of the 1024^3 line passes under the TLB / DRAM counters (needs a box whose default is the fast mode)
export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out
(cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -i -E "UTCL|TLB|XNACK|TCC_EA_RDREQ|TCC_HIT|TCC_MISS|MALL|TCC_EA_WRREQ_STALL|TCC_TAG_STALL|TCC_EA_RD_STALL" | head -60) > gpurun_out/r05s_counters_avail.txt
head -60 gpurun_out/r05s_counters_avail.txt
for mode in default arena; do
  if [ $mode = arena ]; then export C21CM_ARENA=0; else unset C21CM_ARENA; fi
  timeout 300 python bench.py --hii-dim 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-abi --no-kernel-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode', round(d['ms_per_step'],1))"
  for C in TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_REQUEST TCC_EA_RDREQ_sum TCC_HIT_sum TCC_MISS_sum; do
    (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $REPO/gpurun_out/pmcS_${mode}_$C -o pmc -- \
        python $REPO/bench.py --hii-dim 1024 --steps 1 --warmup 0 --no-cpu-baseline --no-abi --no-kernel-roofline > /dev/null 2>&1)
    f=$(find gpurun_out/pmcS_${mode}_$C -name pmc_counter_collection.csv | head -1)
    [ -n "$f" ] && python - "$f" "$mode" "$C" <<'PY'
import csv, sys, collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"]
    for key in ("line_pass_kernel<1024, 1, 0>","line_pass_kernel<1024, 1, 6>","zw3_ionise_kernel"):
        if key in n: acc[key].append(float(r["Counter_Value"]))
print(sys.argv[2], sys.argv[3], {k: round(sum(v)/len(v),1) for k,v in acc.items()})
PY
    rm -rf gpurun_out/pmcS_${mode}_$C
  done
done
