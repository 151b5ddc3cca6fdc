"""One case of scripts/soak_vjp_tables.py: the PARAMETER gradient of both backward kernels against central differences
of the float64 oracle over a subset of the parameters (which of the two is off when they disagree?).
  python scripts/soak_grad_case.py <seed> <case>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from cnf_ot_amd import FlowConfig, FlowEngine, Params
dev = torch.device("cuda", 0)
cfg = FlowConfig(dim=2); ocfg = oracle.OracleConfig(D=2)
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(want + 1):
  scale = float(rng.choice([0.05, 0.1, 0.2, 0.3, 0.5, 0.8, 1.5]))
  spread = float(rng.choice([1.0, 2.0, 4.0, 6.0]))
  amag = float(10.0 ** rng.uniform(-6, 4))
  w = rng.normal(0, scale, cfg.param_count()).astype(np.float32)
  S, Bs = 4, 9000
  B = S * Bs
  to_base = bool(case & 1)
  pts = rng.normal(0, spread, (B, 2)).astype(np.float32)
  ybar = (rng.normal(0, 1, (B, 2)) * amag).astype(np.float32)
  ldbar = (rng.normal(0, 1, B) * amag).astype(np.float32)
  ts = rng.uniform(0, 1, S).astype(np.float32)
eng = FlowEngine(cfg, dev).load(Params(cfg, torch.from_numpy(w).to(dev)))
t_dev = torch.from_numpy(ts).to(dev)
g = {}
for mode in (0, 2):
  eng.set_pwl(mode)
  gg = torch.zeros(cfg.param_count(), device=dev)
  eng.pass_vjp(torch.from_numpy(pts).to(dev), t_dev if mode == 2 else t_dev.repeat_interleave(Bs)[:, None],
               torch.from_numpy(ybar).to(dev), torch.from_numpy(ldbar).to(dev), to_base, grad=gg)
  g[mode] = gg.cpu().double().numpy()
c_host = np.repeat(ts.astype(np.float64), Bs)
fn = oracle.inverse_logdet if to_base else oracle.forward_logdet
x64, yb, lb = pts.astype(np.float64), ybar.astype(np.float64), ldbar.astype(np.float64)
def F(p):
  y, ld = fn(ocfg, p, x64, c_host)
  return float((y * yb).sum() + (ld * lb).sum())
d = np.abs(g[2] - g[0])
idx = np.concatenate([np.argsort(-d)[:12], np.random.default_rng(0).choice(cfg.param_count(), 12, replace=False)])
scale_g = np.abs(g[0]).max()
print(f"case {want}: scale {scale} spread {spread} adjoint {amag:.1e} to_base {to_base}; |g|inf {scale_g:.4g}; kernels differ by {d.max()/scale_g:.2e}")
w64 = w.astype(np.float64)
for i in idx:
  refs = []
  for h in (1e-6, 3e-6):
    p = w64.copy(); p[i] += h; fp = F(p); p[i] -= 2 * h; refs.append((fp - F(p)) / (2 * h))
  print(f"  param {i:5d}: f64 FD {refs[0]: .6e} (h x3: {refs[1]: .6e})  mlp {g[0][i]: .6e} (err {abs(g[0][i]-refs[0])/scale_g:.1e})  tables {g[2][i]: .6e} (err {abs(g[2][i]-refs[0])/scale_g:.1e})")
