#!/bin/bash
python -m pytest tests/test_kernels_gpu.py -q -x -k "conv3x3" 2>&1 | tail -3
python tools/halo_stamps.py 17 8 16 64 320 320 2>&1 | grep -v amdgpu | head -3
python tools/halo_stamps.py 6 8 16 64 320 320 2>&1 | grep -v amdgpu | head -3
python tools/halo_stamps.py 17 8 4 16 1280 1280 2>&1 | grep -v amdgpu | head -2
bash tools/ab2.sh "pre_pa|DF_LIB_OVERRIDE=ab/libdf_prepa_f16.so" "pa|DF_X=1"
