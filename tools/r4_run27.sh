#!/bin/bash
# producer-specialised tiles: correctness under a timeout first (a hang must not take the box), then the micro-benchmark
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "test_gemm and (18 or 19 or 20)" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "conv3x3 or ln_folded or epilogue_times" 2>&1 | tail -5
timeout 600 python tools/gemm_bench.py "" all 2>&1 | grep -v amdgpu > gpurun_out/p27_bench.txt
python - <<'PY'
import re
for l in open('gpurun_out/p27_bench.txt'):
    name=l.split('|')[0]
    items=re.findall(r'(\S+)/sk(\d+):\s+([\d.]+)us',l)
    best={}
    for t,sk,us in items:
        us=float(us)
        if t not in best or us<best[t][0]: best[t]=(us,sk)
    top=sorted(best.items(),key=lambda kv:kv[1][0])
    print(name, ' '.join(f"{t}/sk{v[1]}:{v[0]:.1f}" for t,v in top[:4]), '|', ' '.join(f"{t}/sk{best[t][1]}:{best[t][0]:.1f}" for t in ('P256x128','P128x128','P2_128x128') if t in best))
PY
