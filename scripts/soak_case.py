"""Detail of one soak case (scripts/soak_pwl.py): the samples where the table path is furthest from the oracle."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import oracle
from oracle import pwl_tables as pt
from cnf_ot_amd import FlowConfig, FlowEngine, Params
dev = torch.device("cuda", 0)
cfg = FlowConfig(dim=2); ocfg = oracle.OracleConfig(D=2)
seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(target + 1):
  scale = float(rng.choice([0.05, 0.1, 0.2, 0.3, 0.5, 0.8, 1.5]))
  w = rng.normal(0, scale, 1200)
  kind = case % 6
  if kind == 1:
    w[rng.random(1200) < 0.5] = 0.0
  elif kind == 2:
    for l in range(2):
      blk = w[16 + 592 * l: 16 + 592 * (l + 1)]
      blk[16:24] = blk[24:32]; blk[0:8] = blk[8:16]; blk[32:40] = blk[40:48]
  elif kind == 3:
    w[:16] *= 0.01
    for l in range(2):
      w[16 + 592 * l + 320: 16 + 592 * (l + 1)] *= 0.05
  elif kind == 4:
    for l in range(2):
      w[16 + 592 * l + 32: 16 + 592 * l + 48] *= 40.0
  params = w.astype(np.float32)
  S, Bs = 4, 4000
  ts = rng.uniform(-0.5, 1.5, S).astype(np.float32)
  spread = float(rng.choice([1.0, 3.0, 8.0]))
  noise = (rng.normal(size=(S * Bs, 2)) * spread).astype(np.float32)
eng = FlowEngine(cfg, dev).load(Params(cfg, torch.from_numpy(params).to(dev)))
y64, lp64 = oracle.sample_logprob(ocfg, params.astype(np.float64), noise.astype(np.float64), np.repeat(ts.astype(np.float64), Bs))
out = {}
for mode in (0, 2):
  eng.set_pwl(mode)
  y, lp = eng.sample_logprob(torch.from_numpy(noise).to(dev), torch.from_numpy(ts).to(dev))
  out[mode] = y.cpu().double().numpy()
e0 = np.abs(out[0] - y64); e2 = np.abs(out[2] - y64)
idx = np.argsort(-e2.max(1))[:12]
for l in range(2):
  for s in range(S):
    n = pt.build_table(params[16 + 592 * l: 16 + 592 * (l + 1)].astype(np.float64), float(ts[s]))[0]
    print(f"layer {l} slice {s} c={ts[s]:.3f}: {n.size} breakpoints, range [{n.min() if n.size else 0:.2f}, {n.max() if n.size else 0:.2f}]")
for i in idx:
  print(f"sample {i} slice {i // Bs} noise {noise[i]} y64 {y64[i]} y_mlp {out[0][i]} y_tables {out[2][i]} err mlp {e0[i]} err tables {e2[i]}")
