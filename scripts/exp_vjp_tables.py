"""cnf_pass_vjp at dim 2: the table form (per-piece sufficient statistics) vs the MLP backward, one pass over
32 slices x 131 072 points (config 5's per-GPU share) and 32 x 4 096 (config 3's time batch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnf_ot_amd import FlowConfig, FlowEngine, Params
dev = torch.device("cuda", 0)
cfg = FlowConfig(dim=2)
params = Params.random(cfg, 0.2, seed=5, device=dev)
eng = FlowEngine(cfg, dev).load(params)
for S, Bs in ((32, 131072), (32, 4096), (2, 2097152)):
  B = S * Bs
  pts = torch.randn(B, 2, device=dev); ybar = torch.randn(B, 2, device=dev); ldbar = torch.randn(B, device=dev)
  ts = torch.linspace(0.05, 0.95, S, device=dev)
  for to_base in (False, True):
    for mode, label in ((0, "mlp"), (2, "tables")):
      eng.set_pwl(mode)
      g = torch.zeros(cfg.param_count(), device=dev)
      for want_x in (False,):
        eng.pass_vjp(pts, ts, ybar, ldbar, to_base, grad=g, want_xbar=want_x); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): eng.pass_vjp(pts, ts, ybar, ldbar, to_base, grad=g, want_xbar=want_x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"{S:3d} x {Bs:8d} to_base={to_base!s:5} {label:7s} path={eng.last_path():7s}: {dt*1e3:8.3f} ms  ({B/dt/1e9:.2f} G passes/s)", flush=True)
