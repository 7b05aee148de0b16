set -x
mkdir -p gpurun_out/r3m
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r3m/pytest.txt
bash tools/ab.sh "all|DF_X=1" "nooutdot|DF_NO_OUTDOT=1" "xostore|DF_XO_STORE=1" > gpurun_out/r3m/ab.txt 2>&1
