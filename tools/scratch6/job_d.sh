export TMPDIR=/tmp
REPO=$PWD
for src in 0 1; do
(cd /tmp && PYTHONPATH=$REPO timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_tab$src -o tab -- python $REPO/tools/time_abi_ionize.py 512 $src 9.0 > /dev/null 2>&1)
F=$(find gpurun_out/prof_tab$src -name "*kernel_stats.csv" | head -1)
echo "== source model $src"; python tools/kernel_stats_brief.py $F 12
cp $F gpurun_out/r06_abi_table_src${src}_kernel_stats.csv
find gpurun_out/prof_tab$src -name "*.csv" -size +2M -delete
done
