# placement walk: hipMalloc chunks vs VMM chunks (first call cost, steady state, trace)
for alloc in vmm malloc vmm vmm; do
  C21CM_WS_PLACE_ALLOC=$alloc C21CM_WS_TRACE=1 python bench.py --no-cpu-baseline --no-abi --steps 20 --warmup 3 > gpurun_out/pl_$alloc.json 2> gpurun_out/pl_$alloc.err
  echo "== $alloc"; grep "\[place\]" gpurun_out/pl_$alloc.err | head -40
  python - <<PY
import json
d=json.load(open("gpurun_out/pl_$alloc.json"))
print("$alloc", "ms", round(d["ms_per_step"],3), "first_call_ms", round(d["first_call_ms"],1), d["placement"])
PY
done
# 1024^3
for alloc in malloc vmm; do
  C21CM_WS_PLACE_ALLOC=$alloc C21CM_WS_TRACE=1 python bench.py --hii-dim 1024 --no-cpu-baseline --no-abi --steps 3 --warmup 1 --no-kernel-roofline > gpurun_out/pl1024_$alloc.json 2> gpurun_out/pl1024_$alloc.err
  grep "\[place\]" gpurun_out/pl1024_$alloc.err | head -30
  python - <<PY
import json
d=json.load(open("gpurun_out/pl1024_$alloc.json"))
print("1024 $alloc", "ms", round(d["ms_per_step"],3), "first_call_ms", round(d["first_call_ms"],1), d["placement"])
PY
done
