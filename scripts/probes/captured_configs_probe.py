"""CapturedUpdate on the BASELINE training configs' per-GPU shapes: does the capture go through, what does a replayed
step cost next to the eager one."""
import sys, os, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cnf_ot_amd import solvers
CASES = {
  "cfg3": ({"general": {"type": "rwpo", "dim": 2, "t_batch_size": 32}, "rwpo": {"T": 1, "beta": 1, "a": 1, "pot_type": "quadratic"}}, 131072),
  "cfg4": ({"general": {"type": "fp", "dim": 10, "t_batch_size": 32}, "fp": {"T": 1, "a": 1, "sigma": 0.5, "velocity_field_type": "ou"}}, 32768),
  "cfg5": ({"general": {"type": "ot", "dim": 2, "t_batch_size": 32}, "ot": {"subtype": "obstacle"}}, 4194304),
}
for name, (ov, B) in CASES.items():
  ov = dict(ov); ov["train"] = {"batch_size": B, "lr": 1e-3}
  config = solvers.load_config(overrides=ov)
  for replay in (False, True):
    try:
      model = solvers.build_model(config); params = model.init(7)
      opt = solvers.Adam(1e-3); st = opt.init(params)
      upd = solvers.CapturedUpdate(solvers.bind_loss(config, model), opt, B, replay=replay)
      for step in range(4): upd(params, 100 + step, 5000.0, st)
      torch.cuda.synchronize(); t0 = time.perf_counter()
      for step in range(20): loss, _, _ = upd(params, 200 + step, 5000.0, st)
      torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
      print(f"{name} B={B} replay={replay}: {dt * 1e3:.3f} ms per step, loss {float(loss):.4f}", flush=True)
    except Exception:
      print(f"{name} B={B} replay={replay}: FAILED", flush=True)
      traceback.print_exc(limit=8)
