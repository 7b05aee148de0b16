set -x
mkdir -p gpurun_out/r3f
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "upsample_conv_phase" 2>&1 | tail -15 > gpurun_out/r3f/ups_tests.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3f/pytest.txt
bash tools/ab.sh "ups4|DF_X=1" "noups4|DF_NO_UPS4=1" > gpurun_out/r3f/ab.txt 2>&1
python bench.py --steps 25 --warmup 5 --no-cpu-baseline --dump-ops gpurun_out/r3f/ops.csv > gpurun_out/r3f/bench.json 2> gpurun_out/r3f/bench.err
python tools/vae_bench.py 10 > gpurun_out/r3f/vae.txt 2>&1
DF_NO_UPS4=1 python tools/vae_bench.py 10 > gpurun_out/r3f/vae_noups4.txt 2>&1
