#!/bin/bash
# two-line recombination barrier kernel with the lean loads (N_rec rows parked in LDS, 16-byte mask rows): A/B
for v in default rclean default rclean; do
  lib=$PWD/variants/$v/lib21cmfast_hip.so; [ $v = default ] && lib=$PWD/21cmfast_amd/lib21cmfast_hip.so
  echo "== $v"
  C21CM_LIB=$lib python tools/time_recomb.py 512 3 inhomogeneous_cell 2>&1 | tail -1
  C21CM_LIB=$lib python tools/time_recomb.py 512 3 homogeneous 2>&1 | tail -1
done
