timeout 600 python -m pytest tests/test_gpu_recomb.py -m gpu -x -q -k "fused_recombination_loop_equals" 2>&1 | tail -4
timeout 300 python tools/time_recomb.py 512 > gpurun_out/recomb_timing.json 2> gpurun_out/recomb_timing.err; cat gpurun_out/recomb_timing.json; tail -3 gpurun_out/recomb_timing.err
