for i in 1 2; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-abi > gpurun_out/m16.json 2> gpurun_out/m16.err; python -c "import json;d=json.load(open('gpurun_out/m16.json'));print('mask16 ', d['ms_per_step'], [ (k['kernel'][:18],round(k['ms'],4)) for k in d['roofline']['other_kernels']])"
C21CM_LIB=variants/mask2/lib21cmfast_hip.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-abi > gpurun_out/m2.json 2> gpurun_out/m2.err; python -c "import json;d=json.load(open('gpurun_out/m2.json'));print('mask2  ', d['ms_per_step'], [ (k['kernel'][:18],round(k['ms'],4)) for k in d['roofline']['other_kernels']])"
done
timeout 600 python -m pytest tests/test_gpu_recomb.py tests/test_gpu_ionize.py -m gpu -x -q -k "fused or two_radii or parity or full_size" 2>&1 | tail -4
