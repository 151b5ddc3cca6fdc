"""Eager vs replayed CapturedUpdate at a table-backward batch size: where do the parameters part, and by how much?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cnf_ot_amd import solvers
dev = torch.device("cuda", 0)
B = 1 << 21
tbs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ov = {"general": {"type": "ot", "t_batch_size": tbs}, "train": {"batch_size": B, "lr": 1e-3}}
config = solvers.load_config(overrides=ov)
hist = []
for replay in (False, True, False):
  model = solvers.build_model(config)
  params = model.init(7)
  params.flat.add_(0.05 * torch.randn(params.flat.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1)))
  opt = solvers.Adam(1e-3); st = opt.init(params)
  upd = solvers.CapturedUpdate(solvers.bind_loss(config, model), opt, B, replay=replay)
  h = []
  for step in range(6):
    loss, params, st = upd(params, 1000 + 17 * step, 5000.0, st)
    h.append((params.flat.clone(), loss.clone()))
  torch.cuda.synchronize()
  hist.append(h)
for name, (a, b) in (("eager vs replay", (hist[0], hist[1])), ("eager vs eager", (hist[0], hist[2]))):
  for step, ((pa, la), (pb, lb)) in enumerate(zip(a, b)):
    print(name, "step", step, "params max|d|", (pa - pb).abs().max().item(), "loss", la.item(), lb.item())
