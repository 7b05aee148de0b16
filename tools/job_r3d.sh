set -x
mkdir -p gpurun_out/r3d
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3d/pytest.txt
python -m pytest tests/test_multi_rank_gpu.py tests/test_vocoder_gpu.py tests/test_configs_gpu.py -m gpu -q -s 2>&1 | grep -v "^$" | grep -i "rank\|nnls\|configs\|candidate\|passed\|failed\|Error" | cut -c1-600 > gpurun_out/r3d/verbose.log
timeout 900 python tools/chk_probe.py 150 --partner --mode hash --out gpurun_out/r3d/hash_partner.json > gpurun_out/r3d/hash_partner.log 2>&1
bash tools/ab.sh "r3d|DF_X=1" "r2|DF_LIB_OVERRIDE=$PWD/ab/libdf_r2_f16.so" > gpurun_out/r3d/ab.txt 2>&1
