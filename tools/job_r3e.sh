set -x
mkdir -p gpurun_out/r3e
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "groupnorm_from_producer" 2>&1 | tail -15 > gpurun_out/r3e/gn_tests.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3e/pytest.txt
bash tools/ab.sh "gnstats|DF_X=1" "nognstats|DF_NO_GNSTATS=1" > gpurun_out/r3e/ab.txt 2>&1
python bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-modes --no-vae --dump-ops gpurun_out/r3e/ops.csv > gpurun_out/r3e/bench.json 2> gpurun_out/r3e/bench.err
