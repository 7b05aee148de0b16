set -x
mkdir -p gpurun_out/r3b
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3b/pytest.txt
python -m pytest tests/test_multi_rank_gpu.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -8 > gpurun_out/r3b/two_rank.log
timeout 900 python tools/chk_probe.py 150 --partner --mode hash --out gpurun_out/r3b/hash_partner.json > gpurun_out/r3b/hash_partner.log 2>&1
timeout 900 python tools/chk_probe.py 60 --partner --out gpurun_out/r3b/chk_partner.json > gpurun_out/r3b/chk_partner.log 2>&1
echo "--- new lib conv bench" > gpurun_out/r3b/gemm_bench.txt
python tools/gemm_bench.py conv >> gpurun_out/r3b/gemm_bench.txt 2>&1
echo "--- r2 lib conv bench" >> gpurun_out/r3b/gemm_bench.txt
DF_LIB_OVERRIDE=$PWD/ab/libdf_r2.so python tools/gemm_bench.py conv >> gpurun_out/r3b/gemm_bench.txt 2>&1
echo "--- new lib lin bench" >> gpurun_out/r3b/gemm_bench.txt
python tools/gemm_bench.py lin >> gpurun_out/r3b/gemm_bench.txt 2>&1
echo "--- r2 lib lin bench" >> gpurun_out/r3b/gemm_bench.txt
DF_LIB_OVERRIDE=$PWD/ab/libdf_r2.so python tools/gemm_bench.py lin >> gpurun_out/r3b/gemm_bench.txt 2>&1
bash tools/ab.sh "r3b|DF_X=1" "r2|DF_LIB_OVERRIDE=$PWD/ab/libdf_r2_f16.so" > gpurun_out/r3b/ab.txt 2>&1
