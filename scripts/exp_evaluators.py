"""The reference's two Monte-Carlo evaluators at their full shape (cnf_ot/utils.py:311-389: 10 000 time-slices x
65 536 samples, dim 2): wall time of one call, table kernels (default) vs the per-sample MLP kernels."""
import sys, time
sys.path.insert(0, ".")
import torch
from cnf_ot_amd import RQSFlow, Params, utils
dev = torch.device("cuda", 0)
model = RQSFlow(event_shape=(2,), num_layers=2, hidden_sizes=[16, 16], num_bins=5)
params = Params.random(model.cfg, 0.2, seed=42, device=dev)
for name, fn in (("calc_kinetic_energy", lambda: utils.calc_kinetic_energy(model.apply.sample, params, 7, dim=2)),
                 ("calc_score_kinetic_energy", lambda: utils.calc_score_kinetic_energy(model.apply.sample, model.apply.log_prob, params, dim=2, rng=7))):
  for mode, label in ((1, "tables"), (0, "mlp")):
    model.terms_backend(params).set_pwl(mode)
    v = fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): v = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    passes = 10000 * 65536 * (2 if "score" not in name else 7)
    print(f"{name:28s} {label:7s}: {dt*1e3:8.2f} ms per call  ({passes/dt/1e9:.1f} G flow passes/s)  value {float(v):.6f}", flush=True)
