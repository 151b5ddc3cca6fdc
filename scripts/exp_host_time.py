"""Host-side cost of one value_and_grad call (time until the call returns, nothing waited for) next to its GPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnf_ot_amd import applications as app
dev = torch.device("cuda", 0)
for which in sys.argv[1:] or ["cfg3", "cfg4", "cfg5"]:
  step, f, params, B, passes, desc = bench._config_steps(dev, which)
  Bl = B // (8 if which in ("cfg4", "cfg5") else 1)
  vg = app.value_and_grad(f)
  for _ in range(5): vg(params, 42, 5000.0, Bl)
  torch.cuda.synchronize()
  host, total = [], []
  for _ in range(30):
    t0 = time.perf_counter(); vg(params, 42, 5000.0, Bl); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append(t1 - t0); total.append(t2 - t0)
  host.sort(); total.sort()
  print(f"{which}: host enqueue {host[15]*1e3:.3f} ms, call + sync {total[15]*1e3:.3f} ms", flush=True)
