#!/usr/bin/env python3
"""One BASELINE config's loss and value_and_grad, a few times, for rocprofv3 --kernel-trace --stats:
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cfg4 -- python3 scripts/prof_cfg.py cfg4 [reps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cnf_ot_amd import applications as app

which = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
only = sys.argv[3] if len(sys.argv) > 3 else ""   # "vg" / "loss": that one alone (a kernel trace of just it)
dev = torch.device("cuda", 0)
step, f, params, B, passes, desc = bench._config_steps(dev, which)
share = 8 if which in ("cfg4", "cfg5") else 1
Bl = B // share
vg = app.value_and_grad(f)
for fn, name in ((lambda: f(params, 42, 5000.0, Bl), "loss"), (lambda: vg(params, 42, 5000.0, Bl), "value_and_grad")):
  if (only == "vg" and name == "loss") or (only == "loss" and name != "loss"):
    continue
  fn(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps):
    fn()
  torch.cuda.synchronize()
  print(f"{which} {name}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per call (B on this GPU {Bl})", flush=True)
