"""One case of scripts/soak_vjp_tables.py in detail: the samples where the two backward kernels disagree most,
against central differences of the float64 oracle (is it the tables, the MLP kernel, or float32 itself?)."""
import sys
sys.path.insert(0, ".")
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
params = Params(cfg, torch.from_numpy(w).to(dev))
eng = FlowEngine(cfg, dev).load(params)
t_dev = torch.from_numpy(ts).to(dev)
out = {}
for mode in (0, 2):
  eng.set_pwl(mode)
  g = torch.zeros(cfg.param_count(), device=dev)
  xb = eng.pass_vjp(torch.from_numpy(pts).to(dev), t_dev if mode == 2 else t_dev.repeat_interleave(Bs)[:, None],
                    torch.from_numpy(ybar).to(dev), torch.from_numpy(ldbar).to(dev), to_base, grad=g)
  out[mode] = xb.cpu().double().numpy()
d = np.abs(out[2] - out[0]).max(1)
idx = np.argsort(-d)[:6]
c_host = np.repeat(ts.astype(np.float64), Bs)
fn = oracle.inverse_logdet if to_base else oracle.forward_logdet
h = 1e-6
print(f"case {want}: scale {scale} spread {spread} adjoint {amag:.1e} to_base {to_base}; |xbar|inf {np.abs(out[0]).max():.3g}")
for i in idx:
  ref = np.zeros(2)
  for e in range(2):
    xp = pts[i:i + 1].astype(np.float64).copy(); xm = xp.copy(); xp[0, e] += h; xm[0, e] -= h
    yp, lp = fn(ocfg, w.astype(np.float64), xp, c_host[i:i + 1]); ym, lm = fn(ocfg, w.astype(np.float64), xm, c_host[i:i + 1])
    ref[e] = ((yp - ym)[0] @ ybar[i].astype(np.float64) + (lp - lm)[0] * float(ldbar[i])) / (2 * h)
  print(f"  sample {i:6d} x={pts[i]} f64 FD {ref}  mlp {out[0][i]} (err {np.abs(out[0][i]-ref).max():.2e})  tables {out[2][i]} (err {np.abs(out[2][i]-ref).max():.2e})")
