import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, oracle
from cnf_ot_amd import FlowConfig, FlowEngine, Params
dev = torch.device("cuda", 0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = FlowConfig(dim=2, num_layers=L); ocfg = oracle.OracleConfig(D=2, L=L)
rng = np.random.default_rng(5)
S, Bs = 4, 9000; B = S * Bs
w = rng.normal(0, 0.25, FlowConfig(dim=2).param_count()).astype(np.float32)[:cfg.param_count()]
pts = rng.normal(0, 6.0, (B, 2)).astype(np.float32)
ybar = rng.normal(0, 1, (B, 2)).astype(np.float32); ldbar = rng.normal(0, 1, B).astype(np.float32)
ts = rng.uniform(0, 1, S).astype(np.float32); c_host = np.repeat(ts.astype(np.float64), Bs)
w64 = w.astype(np.float64)
def fd(h, yb, lb):
  ref = np.zeros((B, 2))
  for e in range(2):
    xp = pts.astype(np.float64).copy(); xm = xp.copy(); xp[:, e] += h; xm[:, e] -= h
    yp, lp = oracle.forward_logdet(ocfg, w64, xp, c_host); ym, lm = oracle.forward_logdet(ocfg, w64, xm, c_host)
    ref[:, e] = (((yp - ym) * yb).sum(1) + (lp - lm) * lb) / (2 * h)
  return ref
eng = FlowEngine(cfg, dev).load(Params(cfg, torch.from_numpy(w).to(dev))); eng.set_pwl(0)
for name, yb, lb in (("both", ybar, ldbar), ("y only", ybar, 0 * ldbar), ("ld only", 0 * ybar, ldbar)):
  r1, r2 = fd(1e-6, yb, lb), fd(3e-6, yb, lb)
  mag = np.abs(r1).max(1) + 1e-3 * np.median(np.abs(r1).max(1))
  smooth = np.abs(r1 - r2).max(1) <= 1e-3 * mag
  g = torch.zeros(cfg.param_count(), device=dev)
  xb = eng.pass_vjp(torch.from_numpy(pts).to(dev), torch.from_numpy(ts).to(dev).repeat_interleave(Bs)[:, None],
                    torch.from_numpy(yb).to(dev), torch.from_numpy(lb).to(dev), False, grad=g).cpu().double().numpy()
  rel = np.where(smooth, np.abs(xb - r1).max(1) / mag, 0.0)
  idx = np.argsort(-rel)[:8]
  print(f"== L={L} {name}: beyond 1%: {(rel > 1e-2).sum()}")
  y64, _ = oracle.forward_logdet(ocfg, w64, pts.astype(np.float64), c_host)
  for i in idx:
    print(f"  {i:6d} x={pts[i]} y={y64[i]} ref {r1[i]} got {xb[i]} rel {rel[i]:.2e}")
