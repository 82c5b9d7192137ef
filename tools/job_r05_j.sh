#!/bin/bash
export TMPDIR=/tmp
for p in 1 2 3 4; do ( PYTHONPATH=. timeout 1200 python tools/scratch/diag_1024_race.py > gpurun_out/r05j_diag_$p.txt 2>&1 ) & done
wait
for p in 1 2 3 4; do echo "== proc $p"; grep -v "0 cells differ" gpurun_out/r05j_diag_$p.txt | head -30; grep -c "0 cells differ" gpurun_out/r05j_diag_$p.txt; done
