#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
tools/bin/valu_rate_probe > gpurun_out/r05_valu_rate_probe.txt 2>&1
cat gpurun_out/r05_valu_rate_probe.txt
timeout 2400 python -m pytest tests -x -q -m gpu -n 4 > gpurun_out/r05e_tests.log 2>&1
tail -12 gpurun_out/r05e_tests.log
for acc in fixed double; do echo "CIC_ACC=$acc"; C21CM_CIC_ACC=$acc PYTHONPATH=. timeout 300 python tools/time_cic.py 2>&1 | tail -2; done > gpurun_out/r05e_cic.txt 2>&1
cat gpurun_out/r05e_cic.txt
REPO=$PWD
(cd /tmp && PYTHONPATH=$REPO timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_abi0 -o abi0 -- python $REPO/tools/time_abi_ionize.py 512 0 9.0 > /dev/null 2>&1)
python tools/kernel_stats_brief.py $(find gpurun_out/prof_abi0 -name "*kernel_stats.csv" | head -1) 16 > gpurun_out/r05e_abi_const_ion_eff_kernel_stats_brief.txt 2>&1
cat gpurun_out/r05e_abi_const_ion_eff_kernel_stats_brief.txt
