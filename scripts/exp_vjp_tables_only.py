"""The table form of cnf_pass_vjp alone (both directions), config 5's shapes: 32 x 131 072 and 1 x 4 194 304 points."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnf_ot_amd import FlowConfig, FlowEngine, Params
dev = torch.device("cuda", 0)
cfg = FlowConfig(dim=2)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
params = Params.random(cfg, scale, seed=5, device=dev)
eng = FlowEngine(cfg, dev).load(params)
eng.set_pwl(2)
for S, Bs in ((32, 131072), (1, 4194304)):
  B = S * Bs
  pts = torch.randn(B, 2, device=dev); ybar = torch.randn(B, 2, device=dev); ldbar = torch.randn(B, device=dev)
  ts = torch.linspace(0.05, 0.95, S, device=dev)
  for to_base in (False, True):
    g = torch.zeros(cfg.param_count(), device=dev)
    eng.pass_vjp(pts, ts, ybar, ldbar, to_base, grad=g, want_xbar=False); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): eng.pass_vjp(pts, ts, ybar, ldbar, to_base, grad=g, want_xbar=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"{S:3d} x {Bs:8d} to_base={to_base!s:5} path={eng.last_path():7s}: {dt*1e3:8.3f} ms  ({B/dt/1e9:.2f} G passes/s)", flush=True)
