#!/bin/bash
python -m pytest tests/test_kernels_gpu.py -q -x -k "conv3x3" 2>&1 | tail -3
python tools/cold_probe.py conv 5,6,7,17,23 2>/dev/null
bash tools/ab2.sh "head|DF_X=1" "no_h4w|DF_TILE_CAP=23"
