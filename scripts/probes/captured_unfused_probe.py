"""CapturedUpdate where the score terms run unfused (dim >= 6: applications._score_terms_unfused inside the graph):
the device-keyed body eagerly, then replayed -- prints the exception if the capture refuses an operation."""
import sys, os, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cnf_ot_amd import solvers
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 6
kind = sys.argv[2] if len(sys.argv) > 2 else "fp"
ov = {"general": {"type": kind, "dim": dim, "t_batch_size": 2}, "fp": {"velocity_field_type": "ou"},
      "train": {"batch_size": 2048, "lr": 1e-3}}
config = solvers.load_config(overrides=ov)
res = []
for replay in (False, True):
  try:
    model = solvers.build_model(config)
    params = model.init(7)
    opt = solvers.Adam(1e-3); st = opt.init(params)
    upd = solvers.CapturedUpdate(solvers.bind_loss(config, model), opt, 2048, replay=replay)
    for step in range(6):
      loss, params, st = upd(params, 1000 + 17 * step, 5000.0, st)
    torch.cuda.synchronize()
    res.append(params.flat.clone())
    print(f"{kind} dim {dim} replay={replay}: ok, loss {float(loss):.4f}", flush=True)
  except Exception:
    print(f"{kind} dim {dim} replay={replay}: FAILED", flush=True)
    traceback.print_exc(limit=6)
    break
if len(res) == 2:
  print("eager vs replayed parameters: max |d| =", (res[0] - res[1]).abs().max().item())
