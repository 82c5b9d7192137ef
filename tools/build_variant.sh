#!/bin/bash
# Build lib21cmfast_hip.so with extra -D switches into variants/<name>/ (A/B kernel experiments;
# run with C21CM_LIB=variants/<name>/lib21cmfast_hip.so).  usage: tools/build_variant.sh name "-DFOO=1 ..."
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/variants/$name
mkdir -p $out/hip $out/host
cd $root/21cmfast_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-array-bounds -I../../include -Ihip -Ihost $*"
for f in hip/*.hip; do
  o=$out/hip/$(basename ${f%.hip}).o
  # only fft_native / ionize kernels see the experiment switches; reuse the in-tree objects otherwise
  if grep -q "C21X_" $f; then /opt/rocm/bin/hipcc $FLAGS -c $f -o $o & else cp ${f%.hip}.o $o; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/lib21cmfast_hip.so $out/hip/*.o host/*.o -L/opt/rocm/lib -lrocfft -lgomp -lm -ldl -Wl,-rpath,/opt/rocm/lib -Wl,-Bsymbolic-functions
rm -rf $out/hip $out/host
echo built $out/lib21cmfast_hip.so
