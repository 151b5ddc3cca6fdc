"""Randomised soak of the backward kernels' INPUT ADJOINTS against the float64 oracle: for every sample,
xbar = d(y . ybar + logdet * ldbar) / dx by central differences of the oracle's flow pass (two step sizes: samples where
they disagree sit on a ReLU kink or a spline knot of the float64 function and are not counted).  MLP backward and table
backward, both directions, parameter scales 0.05 .. 1.5, input spreads 1 .. 6.
  python scripts/soak_xbar_oracle.py <seed> <cases> [layers]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from cnf_ot_amd import FlowConfig, FlowEngine, Params
dev = torch.device("cuda", 0)
L = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cfg = FlowConfig(dim=2, num_layers=L); ocfg = oracle.OracleConfig(D=2, L=L)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
S, Bs = 4, 9000
B = S * Bs
bad_total = 0
for case in range(n_cases):
  scale = float(rng.choice([0.05, 0.1, 0.2, 0.3, 0.5, 0.8, 1.5]))
  spread = float(rng.choice([1.0, 2.0, 4.0, 6.0]))
  to_base = bool(case & 1)
  w = rng.normal(0, scale, cfg.param_count()).astype(np.float32)
  pts = rng.normal(0, spread, (B, 2)).astype(np.float32)
  ybar = rng.normal(0, 1, (B, 2)).astype(np.float32)
  ldbar = rng.normal(0, 1, B).astype(np.float32)
  ts = rng.uniform(0, 1, S).astype(np.float32)
  c_host = np.repeat(ts.astype(np.float64), Bs)
  fn = oracle.inverse_logdet if to_base else oracle.forward_logdet
  w64 = w.astype(np.float64)
  def fd(h):
    ref = np.zeros((B, 2))
    for e in range(2):
      xp = pts.astype(np.float64).copy(); xm = xp.copy(); xp[:, e] += h; xm[:, e] -= h
      yp, lp = fn(ocfg, w64, xp, c_host); ym, lm = fn(ocfg, w64, xm, c_host)
      ref[:, e] = (((yp - ym) * ybar).sum(1) + (lp - lm) * ldbar) / (2 * h)
    return ref
  r1, r2 = fd(1e-6), fd(3e-6)
  mag = np.abs(r1).max(1) + 1e-3 * np.median(np.abs(r1).max(1))
  smooth = np.abs(r1 - r2).max(1) <= 1e-3 * mag
  eng = FlowEngine(cfg, dev).load(Params(cfg, torch.from_numpy(w).to(dev)))
  t_dev = torch.from_numpy(ts).to(dev)
  row = []
  for mode, label in ((0, "mlp"), (2, "tables")):
    eng.set_pwl(mode)
    g = torch.zeros(cfg.param_count(), device=dev)
    xb = eng.pass_vjp(torch.from_numpy(pts).to(dev), t_dev if mode == 2 else t_dev.repeat_interleave(Bs)[:, None],
                      torch.from_numpy(ybar).to(dev), torch.from_numpy(ldbar).to(dev), to_base, grad=g).cpu().double().numpy()
    rel = np.abs(xb - r1).max(1) / mag
    rel = np.where(smooth, rel, 0.0)
    n_bad = int((rel > 0.05).sum())           # float32 through an ill-conditioned flow: 5 % of the sample's own adjoint
    bad_total += n_bad
    worst = int(np.argmax(rel))
    row.append(f"{label}: >5% on {n_bad:4d}, median {np.median(rel[smooth]):.1e}, worst {rel[worst]:.2e} (sample {worst}, x={pts[worst]}, ref {r1[worst]}, got {xb[worst]})")
  print(f"case {case:3d} scale {scale:.2f} spread {spread:.0f} to_base {to_base!s:5} smooth {int(smooth.sum())}/{B} | " + " | ".join(row), flush=True)
print(f"cases: {n_cases} samples beyond 5%: {bad_total}")
