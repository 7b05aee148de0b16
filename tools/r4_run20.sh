#!/bin/bash
python -m pytest tests/test_kernels_gpu.py -q -x 2>&1 | tail -3
python -m pytest tests/test_path_gpu.py tests/test_path_fp16_gpu.py tests/test_multi_rank_gpu.py tests/test_configs_gpu.py tests/test_cavp_gpu.py -q -x 2>&1 | tail -3
bash tools/ab2.sh "pre_lean|DF_LIB_OVERRIDE=ab/libdf_prelean_f16.so" "lean_loops|DF_X=1"
python tools/halo_stamps.py 17 8 16 64 320 320 2>&1 | grep -v amdgpu | head -3
