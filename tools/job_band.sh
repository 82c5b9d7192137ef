python -m pytest tests/test_gpu_ionize.py -m gpu -x -q -k "closed_form" 2>&1 | tail -3
for i in 1 2; do
for v in 1 0; do
C21CM_EUL_SUMBAND=$v python bench.py --mode erfc --steps 10 --warmup 3 --no-cpu-baseline --no-abi --no-kernel-roofline > gpurun_out/erfc_band.json 2> gpurun_out/erfc_band.err; python -c "import json;d=json.load(open('gpurun_out/erfc_band.json'));print('sumband $v', d['ms_per_step'], d['roofline']['r_loop']['ms'])"
done; done
