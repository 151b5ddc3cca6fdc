#!/usr/bin/env python3
"""Single-batch latency (the reference's per-call pattern): 65 536 samples per call, 200 calls captured in one HIP
graph and replayed; kernel variants."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnf_ot_amd import FlowConfig, FlowEngine, Params
dev = torch.device("cuda", 0)
cfg = FlowConfig(dim=2)
B = 65536
for label, setup in (("default (mlp, 1 sample/lane)", lambda e: None),
                     ("mlp, 2 samples/lane", lambda e: e.set_samples_per_lane(2)),
                     ("mfma, 1 sample/lane", lambda e: (e.set_mfma(True), e.set_samples_per_lane(1))),
                     ("mfma, 2 samples/lane", lambda e: (e.set_mfma(True), e.set_samples_per_lane(2))),
                     ("tables forced", lambda e: e.set_pwl(2))):
  eng = FlowEngine(cfg, dev).load(Params.random(cfg, 0.2, seed=42, device=dev))
  setup(eng)
  x = eng.normal(1, B); y = torch.empty_like(x); lp = torch.empty(B, device=dev)
  t = torch.linspace(0, 1, 64, device=dev)
  side = torch.cuda.Stream(device=dev)
  side.wait_stream(torch.cuda.current_stream(dev))
  with torch.cuda.stream(side):
    eng.reserve(1)
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g, stream=side):
    for i in range(200):
      eng.sample_logprob(x, t[i % 64:i % 64 + 1], out=y, logp_out=lp)
  g.replay(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(20):
    g.replay()
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 20 / 200
  print(f"{label:32s} path={eng.last_path():7s} {dt*1e6:7.2f} us per 65536-sample call  ({B/dt/1e9:.2f} G samples/s)", flush=True)


# where does the MFMA conditioner stop paying?  kernel time (library HIP events) over launch sizes
print("---- launch-size sweep: kernel us, packed-VALU vs MFMA conditioner (sample_logprob, dim 2) ----")
eng = FlowEngine(cfg, dev).load(Params.random(cfg, 0.2, seed=42, device=dev))
eng.set_pwl(0)
for n in (16384, 32768, 65536, 98304, 131072, 196608, 262144, 524288, 1048576):
  x = eng.normal(1, n); y = torch.empty_like(x); lp = torch.empty(n, device=dev)
  c = torch.tensor([0.5], device=dev)
  row = []
  for mf in (0, 1):
    for spl in (1, 2):
      eng.set_mfma(mf); eng.set_samples_per_lane(spl)
      eng.set_profiling(True)
      for _ in range(30):
        eng.sample_logprob(x, c, out=y, logp_out=lp)
      f_ms, _, launches, _ = eng.read_profile()
      eng.set_profiling(False)
      row.append(f"{'mfma' if mf else 'valu'}{spl}: {f_ms / launches * 1e3:7.2f}")
  print(f"n={n:8d}  " + "  ".join(row), flush=True)
cfg10 = FlowConfig(dim=10)
eng = FlowEngine(cfg10, dev).load(Params.random(cfg10, 0.12, seed=42, device=dev))
print("---- dim 10, log_prob (data -> base, serial in d) ----")
for n in (8192, 32768, 131072, 655360):
  x = eng.normal(1, n); c = torch.tensor([0.5], device=dev)
  row = []
  for mf in (0, 1):
    for spl in (1, 2):
      eng.set_mfma(mf); eng.set_samples_per_lane(spl); eng.set_precise(False)
      eng.set_profiling(True)
      for _ in range(10):
        eng.log_prob(x, c)
      f_ms, _, launches, _ = eng.read_profile()
      eng.set_profiling(False)
      row.append(f"{'mfma' if mf else 'valu'}{spl}: {f_ms / launches * 1e3:8.2f}")
  print(f"n={n:8d}  " + "  ".join(row), flush=True)
