"""One training step at the reference's default size (config/mfc.yaml:35-40: batch 2 048), OT: eager `update`
(host draws), the device-keyed body run eagerly, and the same body replayed from ONE HIP graph (solvers.CapturedUpdate)."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnf_ot_amd import solvers
config = solvers.load_config(overrides={"general": {"type": "ot", "t_batch_size": 1}})


def run(name, make, n=400):
  m = solvers.build_model(config); p = m.init(1); opt = solvers.Adam(1e-3); st = opt.init(p)
  upd = make(solvers.bind_loss(config, m), opt)
  for i in range(20): upd(p, i + 1, 5000.0, st)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for i in range(n): upd(p, i + 100, 5000.0, st)
  torch.cuda.synchronize()
  print(f"{name}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms per step (async issue + final sync)", flush=True)


run("eager update (host draws)", lambda f, o: solvers.make_update(f, o, 2048))
run("device-keyed body, eager", lambda f, o: solvers.CapturedUpdate(f, o, 2048, replay=False))
run("device-keyed body, ONE HIP graph replayed", lambda f, o: solvers.CapturedUpdate(f, o, 2048))
