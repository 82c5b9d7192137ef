cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ic -o ic -- env PYTHONPATH=$GRAFT_REPO_ROOT python $GRAFT_REPO_ROOT/tools/time_ic_pf.py 256 512 > $GRAFT_REPO_ROOT/gpurun_out/ic_pf.json 2> $GRAFT_REPO_ROOT/gpurun_out/ic_prof.err
cd $GRAFT_REPO_ROOT
python tools/kernel_stats_brief.py $(find gpurun_out/prof_ic -name "*kernel_stats.csv" | head -1) 30
cat gpurun_out/ic_pf.json
