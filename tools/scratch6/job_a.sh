set -x
python -m pytest tests/test_gpu_shard_shim.py tests/test_gpu_placement.py "tests/test_gpu_ionize.py::test_config3_full_size_vs_oracle" -q -x 2>&1 | tail -40 > gpurun_out/r06_a_tests.log
python bench.py > gpurun_out/r06_a_bench.json 2> gpurun_out/r06_a_bench.err
tail -5 gpurun_out/r06_a_bench.err
