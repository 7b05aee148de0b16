set -x
mkdir -p gpurun_out/r3g
python -m pytest tests/test_kernels_gpu.py tests/test_path_gpu.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r3g/pytest.txt
bash tools/ab.sh "attnfma|DF_X=1" > gpurun_out/r3g/ab.txt 2>&1
bash tools/profile_round.sh > gpurun_out/r3g/profile_round.log 2>&1
bash tools/sq_counters.sh gpurun_out/r3g/sq > gpurun_out/r3g/sq.log 2>&1
