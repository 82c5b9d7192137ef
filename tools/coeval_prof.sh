cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_coeval -o c -- env PYTHONPATH=$GRAFT_REPO_ROOT python $GRAFT_REPO_ROOT/tools/time_coeval.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/kernel_stats_brief.py $(find gpurun_out/prof_coeval -name "*kernel_stats.csv" | head -1) 22
