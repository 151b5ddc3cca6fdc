import sys, time, cProfile, pstats, io
sys.path.insert(0, ".")
import torch
from cnf_ot_amd import solvers
config = solvers.load_config(overrides={"general": {"type": "ot", "t_batch_size": 1}})
m = solvers.build_model(config); p = m.init(1); opt = solvers.Adam(1e-3); st = opt.init(p)
upd = solvers.make_update(solvers.bind_loss(config, m), opt, 2048)
for i in range(20): upd(p, i + 1, 5000.0, st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200): upd(p, i + 100, 5000.0, st)
torch.cuda.synchronize()
print("step ms (async issue + final sync):", (time.perf_counter() - t0) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable()
for i in range(200): upd(p, i + 1000, 5000.0, st)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
