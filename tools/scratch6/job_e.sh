for i in 1 2; do for p in 1; do echo "xe pair $p"; C21CM_EUL_PAIR=$p PYTHONPATH=. python tools/time_abi_ionize.py 512 1 9.0 1 2>&1 | tail -1; done; done
C21CM_WS_PLACE=0 PYTHONPATH=. python tools/time_abi_ionize.py 512 1 9.0 1 2>&1 | tail -1
export TMPDIR=/tmp; REPO=$PWD
(cd /tmp && PYTHONPATH=$REPO timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_xe -o xe -- python $REPO/tools/time_abi_ionize.py 512 1 9.0 1 > /dev/null 2>&1)
F=$(find gpurun_out/prof_xe -name "*kernel_stats.csv" | head -1); python tools/kernel_stats_brief.py $F 10; cp $F gpurun_out/r06_abi_table_xe_kernel_stats.csv
find gpurun_out/prof_xe -name "*.csv" -size +2M -delete
