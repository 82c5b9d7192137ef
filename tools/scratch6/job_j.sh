export PYTHONPATH=.
for v in base skip1 skip3; do
  echo "== $v"
  for i in 1 2; do C21CM_LIB=variants/r6_$v/lib21cmfast_hip.so python tools/time_passes.py 512 2>/dev/null | grep "filters=(0,3)" | grep "kind 8"; done
done
