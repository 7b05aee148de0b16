set -x
mkdir -p gpurun_out/final
bash tools/profile_round.sh > gpurun_out/final/profile_round.log 2>&1
bash tools/sq_counters.sh gpurun_out/final/sq > gpurun_out/final/sq.log 2>&1
python -m pytest tests/test_multi_rank_gpu.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -6 > gpurun_out/final/two_rank_gpu_test.log
DF_DIST_SHARE_GPU0=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/final/bench_2rank_shared_gpu.json 2> gpurun_out/final/bench_2rank.err
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/final/pytest.txt
timeout 600 python tools/chk_probe.py 100 --partner --mode hash --out gpurun_out/final/hash_partner.json > gpurun_out/final/hash_partner.log 2>&1
