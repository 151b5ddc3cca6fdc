#!/usr/bin/env python3
"""Throughput of the fused loss / training kernels on BASELINE.json's other
configurations (parity-test cases, not bench lines): one GPU's share of each.
Writes gpurun_out/configs.json.  Flow passes per sample are counted so that
rates are comparable with bench.py's flow-pass rate."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functools import partial
from cnf_ot_amd import FlowConfig, FlowModel, Params
from cnf_ot_amd import applications as app, utils as amd_utils

dev = torch.device("cuda", 0)
out = {}

def timeit(fn, n=5):
  fn(); torch.cuda.synchronize()
  ts = []
  for _ in range(n):
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
  return float(np.median(ts))

def entry(name, seconds, passes, note):
  out[name] = {"seconds": seconds, "flow_passes": passes, "flow_passes_per_s": passes / seconds, "note": note}
  print(f"{name:28s} {seconds*1e3:9.3f} ms  {passes/seconds/1e9:7.2f} G flow passes/s   {note}")

# config 3: RWPO quadratic T=1 beta=1 dim=2 batch=131072 (reverse-KL + potential on B, kinetic+score on B//32 per slice)
cfg = FlowConfig(dim=2); model = FlowModel(cfg); params = Params.random(cfg, 0.2, seed=42, device=dev)
B, tb = 131072, 32
f = partial(app.rwpo_loss_fn, model, 2, 1.0, 1.0, 0.01, 0.01, tb, "quadratic", 1.0)
passes = 2 * B + tb * (B // 32) * (3 + 4)
entry("cfg3_rwpo_loss", timeit(lambda: f(params, 42, 5000.0, B)), passes, f"loss only, t_batch_size={tb}")
vg = app.value_and_grad(f)
entry("cfg3_rwpo_value_and_grad", timeit(lambda: vg(params, 42, 5000.0, B)), passes, "loss + gradient (forward passes counted)")

# config 4: Fokker-Planck OU a=1 sigma=.5 dim=10, one GPU's 32768-sample shard of batch 262144
cfg10 = FlowConfig(dim=10); model10 = FlowModel(cfg10); params10 = Params.random(cfg10, 0.12, seed=42, device=dev)
B, tb = 32768, 32
f10 = partial(app.fp_loss_fn, model10, 10, 1.0, 1.0, 0.5, 0.01, 0.01, tb, "ou")
passes = B + tb * (B // 32) * (3 + 20)
entry("cfg4_fp_dim10_loss", timeit(lambda: f10(params10, 42, 5000.0, B)), passes, "per-GPU shard 32768, loss only")
vg10 = app.value_and_grad(f10)
entry("cfg4_fp_dim10_value_and_grad", timeit(lambda: vg10(params10, 42, 5000.0, B), n=3), passes, "loss + gradient")

# config 5: OT obstacle dim=2, one GPU's 131072 samples x 32 slices (kinetic + potential per slice on the FULL slice)
be = model.terms_backend(params)
Bs, S = 131072, 32
noise = be.normal(42, Bs)
ts = np.linspace(0, 1, S).astype(np.float32)
kin = app._spec(0, dt=0.01); pot = app._spec(3, subtype=2)
def cfg5():
  be.loss_terms(kin, noise, ts, Bs, True); be.loss_terms(pot, noise, ts, Bs, True)
entry("cfg5_ot_obstacle_slices", timeit(cfg5), Bs * S * 3, "131072 x 32 slices, kinetic + obstacle potential, no [B,D] output")

# the evaluator that defines the benchmark shape (utils.py:311-340) at reduced slice count
S = 256
entry("calc_kinetic_energy_256x65536", timeit(lambda: amd_utils.calc_kinetic_energy(model, params, 1, 65536, S, 2), n=3),
      65536 * S * 2, "256 slices x 65536 incl. Philox noise generation per slice")

# one training step of the reference's default config (batch 2048)
from cnf_ot_amd import solvers
config = solvers.load_config(overrides={"general": {"type": "ot", "t_batch_size": 1}})
m2 = solvers.build_model(config); p2 = m2.init(1); opt = solvers.Adam(1e-3); st = opt.init(p2)
upd = solvers.make_update(solvers.bind_loss(config, m2), opt, 2048)
k = [0]
def step():
  k[0] += 1; upd(p2, k[0], 5000.0, st)
entry("train_step_ot_batch2048", timeit(step, n=20), 2 * 2048 + 64 * 2, "update(): value_and_grad + Adam, reference default config")

os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/configs.json", "w"), indent=1)

# float64 instantiation (exact mode) vs the fp32 default, same 4M samples
be = model.terms_backend(params)
n = 1 << 22
z32 = be.normal(1, n); z64 = z32.double(); c32 = torch.tensor([0.5], device=dev); c64 = c32.double()
y32 = torch.empty_like(z32); l32 = torch.empty(n, device=dev)
t32 = timeit(lambda: be.sample_logprob(z32, c32, out=y32, logp_out=l32))
t64 = timeit(lambda: be.sample_logprob(z64, c64))
entry("sample_logprob_fp32_4M", t32, n, "default kernel")
entry("sample_logprob_fp64_4M", t64, n, "float64 instantiation (exact mode)")
json.dump(out, open("gpurun_out/configs.json", "w"), indent=1)
