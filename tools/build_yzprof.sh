#!/bin/bash
# variants/yzprof/lib21cmfast_hip.so: the library with plane_yz.hip's per-phase tick counters compiled in
# (C21X_YZ_PROF=1), for tools/time_yz.py under C21CM_LIB.  Run after `make`.
set -e
cd "$(dirname "$0")/../21cmfast_amd/csrc"
mkdir -p /tmp/yzp ../../variants/yzprof
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-array-bounds \
  -I../../include -Ihip -Ihost -DC21X_YZ_PROF=1 $EXTRA -c hip/plane_yz.hip -o /tmp/yzp/plane_yz.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/yzprof/lib21cmfast_hip.so \
  $(ls hip/*.o | grep -v plane_yz) /tmp/yzp/plane_yz.o host/*.o -L/opt/rocm/lib -lrocfft -lgomp -lm -ldl \
  -Wl,-rpath,/opt/rocm/lib -Wl,-Bsymbolic-functions
