"""ot_loss_fn value_and_grad (dim 2) with its terms fused (cnf_loss_terms_grad) or composed on the tables, around the
thresholds of applications.TABLE_BACKWARD_MIN_*."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnf_ot_amd import RQSFlow, Params, applications as app
dev = torch.device("cuda", 0)
model = RQSFlow(event_shape=(2,), num_layers=2, hidden_sizes=[16, 16], num_bins=5)
params = Params.random(model.cfg, 0.15, seed=9, device=dev)
for sub in ("free", "obstacle"):
  for B, tbs in ((1 << 17, 32), (1 << 18, 32), (1 << 19, 32), (1 << 20, 32), (1 << 18, 4), (1 << 19, 4)):
    f = lambda p, rng, lam, bs, **kw: app.ot_loss_fn(model, 2, 1.0, 0.01, tbs, sub, p, rng, lam, bs, source="gaussian", **kw)
    vg = app.value_and_grad(f)
    row = []
    for minpts, minslice in ((1 << 40, 1 << 40), (1, 1)):
      app.TABLE_BACKWARD_MIN_POINTS, app.TABLE_BACKWARD_MIN_SLICE = minpts, minslice
      be = model.terms_backend(params)
      be.set_pwl(2 if minpts == 1 else 1)
      for _ in range(3): vg(params, 11, 50.0, B)
      torch.cuda.synchronize(); t0 = time.perf_counter()
      for _ in range(20): vg(params, 11, 50.0, B)
      torch.cuda.synchronize(); row.append((time.perf_counter() - t0) / 20 * 1e3)
      be.set_pwl(1)
    print(f"{sub:8s} B = {B:8d} (sub-batch {B // 32:6d} x {tbs:2d} times): fused {row[0]:.3f} ms, tables {row[1]:.3f} ms", flush=True)
