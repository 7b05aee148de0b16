set -x
mkdir -p gpurun_out/r3a
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3a/pytest.txt
echo "--- new lib conv bench" > gpurun_out/r3a/gemm_bench.txt
python tools/gemm_bench.py conv >> gpurun_out/r3a/gemm_bench.txt 2>&1
echo "--- r2 lib conv bench" >> gpurun_out/r3a/gemm_bench.txt
DF_LIB_OVERRIDE=$PWD/ab/libdf_r2.so python tools/gemm_bench.py conv >> gpurun_out/r3a/gemm_bench.txt 2>&1
echo "--- new lib lin bench" >> gpurun_out/r3a/gemm_bench.txt
python tools/gemm_bench.py lin >> gpurun_out/r3a/gemm_bench.txt 2>&1
timeout 900 python tools/chk_probe.py 60 --partner --mode hash --out gpurun_out/r3a/hash_partner.json > gpurun_out/r3a/hash_partner.log 2>&1
timeout 900 python tools/chk_probe.py 60 --partner --out gpurun_out/r3a/chk_partner.json > gpurun_out/r3a/chk_partner.log 2>&1
timeout 600 python tools/chk_probe.py 30 --out gpurun_out/r3a/chk_alone.json > gpurun_out/r3a/chk_alone.log 2>&1
bash tools/ab.sh "r3a|DF_X=1" "r2|DF_LIB_OVERRIDE=$PWD/ab/libdf_r2_f16.so" > gpurun_out/r3a/ab.txt 2>&1
python bench.py --steps 25 --warmup 5 --dump-ops gpurun_out/r3a/ops.csv > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
