"""The reference's default training run (config/mfc.yaml: OT free, dim 2, batch 2 048, 30 000 steps of update()),
captured and eager: wall time, first / last loss."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cnf_ot_amd import solvers
config = solvers.load_config()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else config["train"]["epochs"]
for capture in (True, False):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  model, params, hist = solvers.train(config, epochs=steps, capture=capture)
  torch.cuda.synchronize(); dt = time.perf_counter() - t0
  h = torch.stack([x.reshape(()) for x in hist]).double().cpu()
  k = max(steps // 100, 1)
  print(f"capture={capture}: {steps} steps in {dt:.2f} s ({dt / steps * 1e3:.4f} ms per step); loss mean of first {k} {h[:k].mean().item():.3f} -> "
        f"last {k} {h[-k:].mean().item():.3f}; finite {torch.isfinite(h).all().item()}", flush=True)
