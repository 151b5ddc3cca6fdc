"""Where the table form of cnf_pass_vjp starts to pay: both forms over pass sizes around the thresholds of cnf_grad.hip
(slices >= 8 192 points, >= 524 288 points per pass)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnf_ot_amd import FlowConfig, FlowEngine, Params
dev = torch.device("cuda", 0)
cfg = FlowConfig(dim=2)
eng = FlowEngine(cfg, dev).load(Params.random(cfg, 0.2, seed=5, device=dev))
for S, Bs in ((1, 65536), (1, 131072), (1, 262144), (1, 524288), (4, 32768), (4, 65536), (8, 65536), (32, 4096), (32, 8192), (32, 16384), (96, 4096), (96, 8192)):
  B = S * Bs
  pts = torch.randn(B, 2, device=dev); ybar = torch.randn(B, 2, device=dev); ldbar = torch.randn(B, device=dev)
  ts = torch.linspace(0.05, 0.95, S, device=dev)
  row = []
  for to_base in (False, True):
    for mode in (0, 2):
      eng.set_pwl(mode)
      g = torch.zeros(cfg.param_count(), device=dev)
      for _ in range(3): eng.pass_vjp(pts, ts, ybar, ldbar, to_base, grad=g, want_xbar=False)
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(20): eng.pass_vjp(pts, ts, ybar, ldbar, to_base, grad=g, want_xbar=False)
      torch.cuda.synchronize()
      row.append((time.perf_counter() - t0) / 20 * 1e3)
  print(f"{S:3d} x {Bs:7d} = {B:8d} points: base->data mlp {row[0]:.3f} tables {row[1]:.3f} | data->base mlp {row[2]:.3f} tables {row[3]:.3f} ms", flush=True)
