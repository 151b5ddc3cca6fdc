#!/usr/bin/env python3
"""Table path vs MLP kernel vs oracle for conditioning inputs far outside the search grid (one flow layer)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import oracle
from cnf_ot_amd import FlowConfig, FlowEngine, Params
dev = torch.device("cuda", 0)
cfg = FlowConfig(dim=2, num_layers=1); ocfg = oracle.OracleConfig(D=2, L=1)
rng = np.random.default_rng(7)
params = rng.normal(0, 0.8, cfg.param_count()).astype(np.float32)
us = np.array([-3e4, -8e3, -1e3, -200, -60, -30, -17, -15, -5, 0, 5, 15, 17, 30, 60, 200, 1e3, 8e3, 3e4], dtype=np.float32)
for v1 in (-12.0, -3.0, 2.0, 12.0):
  x = np.stack([us, np.full_like(us, v1)], 1)
  x = np.repeat(x, 2, 0)                       # pairs (the table kernel has two samples per lane)
  eng = FlowEngine(cfg, dev).load(Params(cfg, torch.from_numpy(params).to(dev)))
  y64, lp64 = oracle.sample_logprob(ocfg, params.astype(np.float64), x.astype(np.float64), [0.3])
  res = {}
  for mode in (0, 2):
    eng.set_pwl(mode)
    y, lp = eng.sample_logprob(torch.from_numpy(x).to(dev), torch.tensor([0.3], device=dev))
    res[mode] = (y.cpu().numpy(), lp.cpu().numpy(), eng.last_path())
  print(f"--- v1 = {v1}  paths {res[0][2]} / {res[2][2]}")
  for i in range(0, len(x), 2):
    print(f"u={x[i,0]:9.1f}  y1: oracle {y64[i,1]:14.6f} mlp {res[0][0][i,1]:14.6f} tables {res[2][0][i,1]:14.6f}   lp: oracle {lp64[i]:12.5f} mlp {res[0][1][i]:12.5f} tables {res[2][1][i]:12.5f}")
