#!/usr/bin/env python3
"""Throughput cost of the precise position path (cnf_model_set_precise) of log_prob: 256 x 65 536 samples,
table path and MLP kernel, precise vs plain fp32; and accuracy on seeds 0..4 vs the float64 oracle."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from cnf_ot_amd import FlowConfig, FlowEngine, Params
dev = torch.device("cuda", 0)
cfg = FlowConfig(dim=2)
eng = FlowEngine(cfg, dev).load(Params.random(cfg, 0.2, seed=42, device=dev))
S, B = 256, 65536
x = eng.normal(1, S * B)
t = torch.linspace(0, 1, S, device=dev)
y, _ = eng.sample_logprob(x, t)
def timeit(fn, n=10):
  fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(n): fn()
  torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for pwl in (1, 0):
  eng.set_pwl(pwl)
  for pr in (True, False):
    eng.set_precise(pr)
    dt = timeit(lambda: eng.log_prob(y, t))
    print(f"log_prob 256x65536  path={eng.last_path():7s} precise={pr}: {dt*1e3:.3f} ms  {S*B/dt/1e9:.1f} G samples/s", flush=True)
  dt = timeit(lambda: eng.sample_logprob(x, t))
  print(f"sample_logprob      path={eng.last_path():7s}: {dt*1e3:.3f} ms  {S*B/dt/1e9:.1f} G samples/s", flush=True)
eng.set_pwl(1); eng.set_precise(True)
ocfg = oracle.OracleConfig(D=2)
for seed in range(5):
  rng = np.random.default_rng(seed)
  p = rng.normal(0, 0.2, 1200).astype(np.float32)
  e = FlowEngine(cfg, dev).load(torch.from_numpy(p).to(dev))
  z = torch.from_numpy(rng.normal(size=(65536, 2)).astype(np.float32)).to(dev)
  c = torch.tensor([0.5], device=dev)
  yy, _ = e.sample_logprob(z, c)
  ref = oracle.log_prob(ocfg, p.astype(np.float64), yy.cpu().double().numpy(), [0.5])
  for mode in (0, 2):
    e.set_pwl(mode)
    for pr in (True, False):
      e.set_precise(pr)
      err = np.abs(e.log_prob(yy, c).cpu().double().numpy() - ref)
      print(f"seed {seed} path={e.last_path():7s} precise={pr}: max {err.max():.2e} p99.9 {np.quantile(err,.999):.2e}", flush=True)
