"""solvers.train at a table-backward batch size (2^21, OT, 8 times), eager and captured: ms per step, the loss curve."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cnf_ot_amd import solvers
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for sub in ("free", "obstacle"):
  ov = {"general": {"type": "ot", "t_batch_size": 8}, "ot": {"subtype": sub}, "train": {"batch_size": 1 << 21, "lr": 1e-3}}
  config = solvers.load_config(overrides=ov)
  for capture in (False, True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model, params, hist = solvers.train(config, epochs=steps, capture=capture)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    h = torch.stack([x.reshape(()) for x in hist]).cpu()
    print(f"{sub} capture={capture}: {dt / steps * 1e3:.3f} ms per step; loss {h[0].item():.1f} -> {h[-1].item():.1f}; finite {torch.isfinite(h).all().item()}; "
          f"path {model.terms_backend(params).last_path()}", flush=True)
