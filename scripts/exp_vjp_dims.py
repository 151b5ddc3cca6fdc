"""cnf_pass_vjp (MLP backward) over the event dimension: time per pass and per conditioner evaluation
(a pass at dim D runs L * (D - 1) conditioners), 655 360 passes = the finite-difference score passes of
config 4's per-GPU share."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnf_ot_amd import FlowConfig, FlowEngine, Params
dev = torch.device("cuda", 0)
B, S = 655360, 32
for D in (2, 3, 4, 6, 10):
  cfg = FlowConfig(dim=D)
  eng = FlowEngine(cfg, dev).load(Params.random(cfg, 0.12, seed=5, device=dev))
  eng.set_pwl(0)
  pts = torch.randn(B, D, device=dev); ybar = torch.randn(B, D, device=dev); ldbar = torch.randn(B, device=dev)
  ts = torch.linspace(0.05, 0.95, S, device=dev)
  for to_base in (False, True):
    for wgrad in (True, False):
      g = torch.zeros(cfg.param_count(), device=dev) if wgrad else None
      eng.pass_vjp(pts, ts, ybar, ldbar, to_base, grad=g, want_xbar=True); torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(5): eng.pass_vjp(pts, ts, ybar, ldbar, to_base, grad=g, want_xbar=True)
      torch.cuda.synchronize()
      dt = (time.perf_counter() - t0) / 5
      nc = cfg.num_layers * (D - 1)
      print(f"D={D:2d} to_base={to_base!s:5} wgrad={wgrad!s:5}: {dt*1e3:7.3f} ms  {dt/B*1e9:6.3f} ns/pass  "
            f"{dt/B/nc*1e9:6.3f} ns/conditioner", flush=True)
