#!/usr/bin/env python3
"""Host-side profile (cProfile) of one BASELINE config's value_and_grad: where the wall time goes when it is
not the kernels.  scripts/exp_host_profile.py cfg5"""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cnf_ot_amd import applications as app
which = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
dev = torch.device("cuda", 0)
step, f, params, B, passes, desc = bench._config_steps(dev, which)
Bl = B // (8 if which in ("cfg4", "cfg5") else 1)
vg = app.value_and_grad(f)
for _ in range(3):
  vg(params, 42, 5000.0, Bl)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(50):
  vg(params, 42, 5000.0, Bl)
torch.cuda.synchronize()
pr.disable()
print(f"{which} value_and_grad: {(time.perf_counter() - t0) / 50 * 1e3:.2f} ms per call")
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
