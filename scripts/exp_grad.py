import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from oracle_backend import OracleBackend
from cnf_ot_amd import _capi, FlowConfig, FlowEngine, Params
from cnf_ot_amd.applications import _spec
dev = torch.device("cuda", 0)
cfg = FlowConfig(dim=2)
params = Params.random(cfg, 0.2, seed=10, device=dev)
eng = FlowEngine(cfg, dev).load(params)
B = 200
pts = eng.normal(11, B)
t = np.array([0.2], dtype=np.float32)
spec = _spec(_capi.TERM_KINETIC, dt=0.01)
grad = torch.zeros(cfg.param_count(), device=dev)
eng.loss_terms_grad(spec, pts, t, B, True, 1.0, grad)
g = grad.cpu().double().numpy()
p64 = params.flat.cpu().double().numpy(); pts64 = pts.cpu()
f = lambda p: float(OracleBackend(cfg, p).loss_terms(spec, pts64, t, B, True).sum())
for idx in (335, 320, 927, 608, 100):
  row = []
  for h in (1e-3, 1e-4, 1e-5, 1e-6, 1e-7):
    p = p64.copy(); p[idx] += h; fp = f(p); p[idx] -= 2*h; fm = f(p)
    row.append((fp - fm) / (2*h))
  print(idx, "gpu", g[idx], "fd(h=1e-3..1e-7)", ["%.5f" % r for r in row])
# per-sample view: which samples dominate d loss / d b1[15] ?
import oracle
from oracle import losses as ol
