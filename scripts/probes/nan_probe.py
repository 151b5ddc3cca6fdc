import sys, torch
sys.path.insert(0, "/root/repo")
from cnf_ot_amd import FlowConfig, FlowEngine, Params
dev = torch.device("cuda", 0)
cfg = FlowConfig(dim=2)
eng = FlowEngine(cfg, dev).load(Params.random(cfg, 0.2, seed=5, device=dev))
B = 1 << 20
x = torch.randn(B, 2, device=dev)
for bad in (float("nan"), float("inf"), -float("inf")):
  x2 = x.clone(); x2[5, 0] = bad; x2[77, 1] = bad
  for mode in (0, 2):
    eng.set_pwl(mode)
    y, lp = eng.sample_logprob(x2, 0.3)
    z, ild = eng.inverse_logdet(x2, 0.3)
    lpd = eng.log_prob(x2, 0.3)
    print(bad, "mode", mode, eng.last_path(), "sample y[5]", y[5].tolist(), "lp[5]", lp[5].item(), "lp[77]", lp[77].item(),
          "| inverse z[5]", z[5].tolist(), "ildj[5]", ild[5].item(), "log_prob[5]", lpd[5].item(), "log_prob[77]", lpd[77].item())
