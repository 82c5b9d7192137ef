#!/bin/bash
# packed complex primitives (in-tree) vs scalar forms (variants/nopk) + rotate item orders; tests on the packed build
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/r05c_ab.txt
run() { # label, env...
  echo "$1" >> gpurun_out/r05c_ab.txt
  shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-abi 2>/dev/null | python tools/bench_brief.py >> gpurun_out/r05c_ab.txt
}
run "nopk" C21CM_LIB=variants/nopk/lib21cmfast_hip.so
run "pk" A=1
run "pk Yrot" C21CM_YORDER=-1
run "pk Xrot Yrot" C21CM_YORDER=-1 C21CM_XORDER=3
run "pk Y1" C21CM_YORDER=1
run "nopk Yrot" C21CM_LIB=variants/nopk/lib21cmfast_hip.so C21CM_YORDER=-1
run "nopk" C21CM_LIB=variants/nopk/lib21cmfast_hip.so
run "pk" A=1
echo "1024:" >> gpurun_out/r05c_ab.txt
for e in "A=1" "C21CM_YORDER=-1 C21CM_XORDER=3" "C21CM_LIB=variants/nopk/lib21cmfast_hip.so"; do
  echo "$e" >> gpurun_out/r05c_ab.txt
  env $e timeout 300 python bench.py --hii-dim 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-abi 2>/dev/null | python tools/bench_brief.py >> gpurun_out/r05c_ab.txt
done
echo "erfc:" >> gpurun_out/r05c_ab.txt
for e in "A=1" "C21CM_LIB=variants/nopk/lib21cmfast_hip.so"; do
  echo "$e" >> gpurun_out/r05c_ab.txt
  env $e timeout 300 python bench.py --mode erfc --no-cpu-baseline --no-abi 2>/dev/null | python tools/bench_brief.py >> gpurun_out/r05c_ab.txt
done
cat gpurun_out/r05c_ab.txt
timeout 2400 python -m pytest tests -x -q -m gpu -n 4 > gpurun_out/r05c_tests.log 2>&1
tail -12 gpurun_out/r05c_tests.log
