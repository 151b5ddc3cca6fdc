"""Interleaved A/B timing of kernel variants in ONE process (methodology rule 24)."""
import sys, time, itertools
import torch
sys.path.insert(0, ".")
from cnf_ot_amd import FlowConfig, FlowEngine, Params, _capi
dev = torch.device("cuda", 0)
cfg = FlowConfig(dim=2)
params = Params.random(cfg, 0.2, seed=42, device=dev)
S, B = 256, 65536
variants = []
for name, mfma, spl in [("valu2", 0, 2), ("mfma2", 1, 2), ("mfma1", 1, 1), ("valu1", 0, 1)]:
  e = FlowEngine(cfg, dev).load(params)
  e.set_mfma(mfma); e.set_samples_per_lane(spl)
  variants.append((name, e))
noise = variants[0][1].normal(42, S * B)
t = torch.linspace(0, 1, S, device=dev)
y = torch.empty(S * B, 2, device=dev); lp = torch.empty(S * B, device=dev)
res = {n: [] for n, _ in variants}
for rnd in range(6):
  for name, e in variants:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
      e.sample_logprob(noise, t, out=y, logp_out=lp)
    e1.record(); torch.cuda.synchronize()
    if rnd > 0: res[name].append(e0.elapsed_time(e1) / 4)
for name, v in res.items():
  v.sort()
  print(f"{name:10s} median {v[len(v)//2]:.4f} ms  min {v[0]:.4f} ms  -> {S*B/v[len(v)//2]/1e6:.2f} G samples/s")
