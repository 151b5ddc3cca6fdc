#!/usr/bin/env python3
"""dim-10 sample_and_log_prob: wave-per-dimension kernel (flow_dpar_kernel) vs the one-sample-per-lane kernel
over batch sizes, kernel time from the library's HIP events."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnf_ot_amd import FlowConfig, FlowEngine, Params
dev = torch.device("cuda", 0)
for D in (10, 3):
  cfg = FlowConfig(dim=D)
  eng = FlowEngine(cfg, dev).load(Params.random(cfg, 0.12, seed=42, device=dev))
  flop = 2 * sum(2 * ((1 + d) * 16 + 256 + 16 * 16) for d in range(1, D)) + 2 * ((D - 1) * 125 + 75)
  for n in (8192, 32768, 131072, 524288, 2097152):
    z = eng.normal(1, n); c = torch.tensor([0.5], device=dev)
    y = torch.empty_like(z); lp = torch.empty(n, device=dev)
    row = []
    for mode, spl in ((0, 0), (2, 1), (2, 2)):
      eng.set_dpar(mode); eng.set_samples_per_lane(spl)
      eng.set_profiling(True)
      for _ in range(12):
        eng.sample_logprob(z, c, out=y, logp_out=lp)
      f_ms, _, launches, _ = eng.read_profile()
      eng.set_profiling(False)
      k = f_ms / launches * 1e-3
      row.append(f"{eng.last_path()}/spl{spl}: {k*1e6:8.1f} us {n/k/1e9:6.2f} G/s frac {flop*n/k/157.3e12:.3f}")
    print(f"D={D} n={n:8d}  " + " | ".join(row), flush=True)
