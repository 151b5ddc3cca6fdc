"""Piecewise-linear conditioner tables (cnf_pwl.h) vs the MLP kernel: parity
against the float64 C oracle and interleaved timing in one process."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import oracle
from cnf_ot_amd import FlowConfig, FlowEngine, Params
dev = torch.device("cuda", 0)
cfg = FlowConfig(dim=2)
params = Params.random(cfg, 0.2, seed=42, device=dev)
ocfg = oracle.OracleConfig(D=2)
pw = params.flat.cpu().double().numpy()
engines = {}
for name, mode in (("mlp", 0), ("pwl", 2)):
  e = FlowEngine(cfg, dev).load(params); e.set_pwl(mode); engines[name] = e

# parity: 5 slices of 10 000 samples (ragged against the 2048 tile), both directions
S, B = 5, 10000
noise = engines["mlp"].normal(7, S * B)
t = torch.linspace(0.05, 0.95, S, device=dev)
c_host = np.repeat(t.cpu().double().numpy(), B)
yo, lpo = oracle.sample_logprob(ocfg, pw, noise.cpu().double().numpy(), c_host)
for name, e in engines.items():
  y, lp = e.sample_logprob(noise, t)
  torch.cuda.synchronize()
  ey = np.abs(y.cpu().double().numpy() - yo).max(); el = np.abs(lp.cpu().double().numpy() - lpo).max()
  lq = e.log_prob(y, t)
  eq = np.abs(lq.cpu().double().numpy() - oracle.log_prob(ocfg, pw, y.cpu().double().numpy(), c_host)).max()
  print(f"[parity {name}] sample err {ey:.2e}  logp err {el:.2e}  log_prob(data->base) err {eq:.2e}")
a = engines["mlp"].sample_logprob(noise, t); b = engines["pwl"].sample_logprob(noise, t)
print("pwl vs mlp: max |dy|", float((a[0] - b[0]).abs().max()), " max |dlp|", float((a[1] - b[1]).abs().max()))

# timing at the bench shape
S, B = 256, 65536
noise = engines["mlp"].normal(42, S * B)
t = torch.linspace(0, 1, S, device=dev)
y = torch.empty(S * B, 2, device=dev); lp = torch.empty(S * B, device=dev)
res = {n: [] for n in engines}
for rnd in range(6):
  for name, e in engines.items():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
      e.sample_logprob(noise, t, out=y, logp_out=lp)
    e1.record(); torch.cuda.synchronize()
    if rnd > 0: res[name].append(e0.elapsed_time(e1) / 4)
for name, v in res.items():
  v.sort()
  print(f"{name:6s} median {v[len(v)//2]:.4f} ms  min {v[0]:.4f} ms  -> {S*B/v[len(v)//2]/1e6:.2f} G samples/s")

# fused loss terms (in-kernel noise): MLP loss kernel vs the tables
from cnf_ot_amd import applications as app, _capi
ts = np.linspace(0, 1, S).astype(np.float32)
specs = {"kinetic": app._spec(_capi.TERM_KINETIC, dt=0.01),
         "kinetic_score": app._spec(_capi.TERM_KINETIC_SCORE, dt=0.01, dx=0.01, coef=0.5),
         "potential": app._spec(_capi.TERM_POTENTIAL, subtype=2),
         "reverse_kl": app._spec(_capi.TERM_REVERSE_KL, T=1.0, beta=4.0)}
for tag, spec in specs.items():
  vals = {}
  for name, e in engines.items():
    for _ in range(2):
      r = e.loss_terms_seeded(spec, 9, ts, B, first_sample=0, slice_stride=B)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
      r = e.loss_terms_seeded(spec, 9, ts, B, first_sample=0, slice_stride=B)
    e1.record(); torch.cuda.synchronize()
    vals[name] = (e0.elapsed_time(e1) / 4, float(r.sum()) / (S * B))
  print(f"[loss {tag}] mlp {vals['mlp'][0]:.3f} ms ({vals['mlp'][1]:.8g})   tables {vals['pwl'][0]:.3f} ms ({vals['pwl'][1]:.8g})")
