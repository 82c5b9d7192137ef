#!/bin/bash
# Hardware-counter study of the pass kernels (diagnostic): one rocprofv3 --pmc run per counter
# group over tools/time_passes.py.   gpurun --timeout 900 -- 'bash tools/pmc_passes.sh'
REPO=$PWD
export TMPDIR=/tmp PYTHONPATH=$REPO
mkdir -p $REPO/gpurun_out/pmc_study
cd /tmp
i=0
for group in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
  "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" \
  "TCC_TAG_STALL_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
  "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT" ; do
  # (a TA_* / GRBM_GUI_ACTIVE group hung the profiler on this pool: every run is under `timeout`)
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_study/g$i -o g -- \
      python $REPO/tools/time_passes.py 512 > /dev/null 2> $REPO/gpurun_out/pmc_study/g$i.err
done
cd $REPO
python - <<'PY' | tee gpurun_out/pmc_study_pass_kernels.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_study/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if any(t in n for t in ("line_pass_kernel", "z_c2r", "zw_", "window_table")):
            key = n.split("(anonymous namespace)::")[1].split("(")[0]
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --pmc study of the pass kernels at 512^3 (tools/pmc_passes.sh over tools/time_passes.py;")
print("# one run per counter group, mean per launch; line_pass<512,1,0> mixes pass X without window and pass Y)")
for k, d in sorted(acc.items()):
    print("==", k)
    m = {}
    for c, v in sorted(d.items()):
        m[c] = sum(v) / len(v)
        print(f"   {c:36s} launches={len(v):3d} mean={m[c]:.4g}")
    try:
        print(f"   -> mean L1->L2 read latency {m['TCP_TCC_READ_REQ_LATENCY_sum'] / m['TCP_TCC_READ_REQ_sum']:.0f} cycles; "
              f"wait-on-memory share of wave cycles {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.2f}; "
              f"LDS-active share {m['SQ_ACTIVE_INST_LDS'] / m['SQ_WAVE_CYCLES']:.3f}; "
              f"VALU-active share {m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES']:.3f}; "
              f"L2 hit rate {m['TCC_HIT_sum'] / max(1.0, m['TCC_REQ_sum']):.2f}")
    except (KeyError, ZeroDivisionError):
        pass
PY
