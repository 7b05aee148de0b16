set -x
mkdir -p gpurun_out/r3h
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r3h/pytest.txt
bash tools/ab.sh "dedup|DF_X=1" "nodedup|DF_NO_CFGDEDUP=1" > gpurun_out/r3h/ab.txt 2>&1
bash tools/sq_counters.sh gpurun_out/r3h/sq > gpurun_out/r3h/sq.log 2>&1
