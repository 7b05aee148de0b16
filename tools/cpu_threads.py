import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd
from diff_foley_amd import synth
import bench
sd = synth.make_state_dict(synth.state_dict_spec(), 0)
for n in (8, 16, 32, 64):
    torch.set_num_threads(n)
    r = bench.cpu_baseline(sd, 3)
    print(n, round(r["value"], 3), flush=True)
