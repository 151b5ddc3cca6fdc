"""Randomised soak of the table path against the MLP kernel: many parameter
sets (scales 0.05 .. 1.5, sparse / duplicated / zeroed weights), conditions and
input ranges.  Reports the worst disagreement relative to the disagreement of
the two kernels with the float64 oracle on the same case."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import oracle
from cnf_ot_amd import FlowConfig, FlowEngine, Params, applications as app, _capi
dev = torch.device("cuda", 0)
L = int(sys.argv[3]) if len(sys.argv) > 3 else 2            # flow layers
cfg = FlowConfig(dim=2, num_layers=L)
ocfg = oracle.OracleConfig(D=2, L=L)
NP = cfg.param_count()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
worst = []
bad = 0
for case in range(n_cases):
  scale = float(rng.choice([0.05, 0.1, 0.2, 0.3, 0.5, 0.8, 1.5]))
  w = rng.normal(0, scale, NP)
  kind = case % 6
  if kind == 1:                      # sparse weights: many exact zeros (a_j = 0, P = 0 branches)
    w[rng.random(NP) < 0.5] = 0.0
  elif kind == 2:                    # duplicated first-layer units: tied breakpoints
    for l in range(L):
      blk = w[16 + 592 * l: 16 + 592 * (l + 1)]
      blk[16:24] = blk[24:32]; blk[0:8] = blk[8:16]; blk[32:40] = blk[40:48]
  elif kind == 3:                    # spline parameters small (near-identity), conditioner large
    w[:16] *= 0.01
    for l in range(L):
      w[16 + 592 * l + 320: 16 + 592 * (l + 1)] *= 0.05
  elif kind == 4:                    # huge first-layer biases: breakpoints far outside the grid
    for l in range(L):
      w[16 + 592 * l + 32: 16 + 592 * l + 48] *= 40.0
  params = w.astype(np.float32)
  S, Bs = 4, 4000
  ts = rng.uniform(-0.5, 1.5, S).astype(np.float32)
  spread = float(rng.choice([1.0, 3.0, 8.0]))
  noise = (rng.normal(size=(S * Bs, 2)) * spread).astype(np.float32)
  eng = FlowEngine(cfg, dev).load(Params(cfg, torch.from_numpy(params).to(dev)))
  out = {}
  for mode in (0, 2):
    eng.set_pwl(mode)
    y, lp = eng.sample_logprob(torch.from_numpy(noise).to(dev), torch.from_numpy(ts).to(dev))
    out[mode] = (y.cpu().double().numpy(), lp.cpu().double().numpy())
  c64 = np.repeat(ts.astype(np.float64), Bs)
  y64, lp64 = oracle.sample_logprob(ocfg, params.astype(np.float64), noise.astype(np.float64), c64)
  # data -> base on the oracle's samples (as float32), and two fused loss terms, on both kernels
  y_in = np.clip(y64, -1e6, 1e6).astype(np.float32)
  lpd64 = oracle.log_prob(ocfg, params.astype(np.float64), y_in.astype(np.float64), c64)
  lpd, terms = {}, {}
  for mode in (0, 2):
    eng.set_pwl(mode)
    lpd[mode] = eng.log_prob(torch.from_numpy(y_in).to(dev), torch.from_numpy(ts).to(dev)).cpu().double().numpy()
    terms[mode] = [eng.loss_terms_seeded(sp, 5, ts, Bs, first_sample=case, slice_stride=Bs).cpu().numpy()
                   for sp in (app._spec(_capi.TERM_KINETIC, dt=0.01), app._spec(_capi.TERM_KINETIC_SCORE, dt=0.01, dx=0.01, coef=0.5),
                              app._spec(_capi.TERM_REVERSE_KL, T=1.0, beta=4.0))]
  fin = np.isfinite(lp64) & np.isfinite(out[0][1]) & np.isfinite(out[2][1])
  e_mlp = np.abs(out[0][1] - lp64)[fin]; e_pwl = np.abs(out[2][1] - lp64)[fin]
  ey_mlp = np.abs(out[0][0] - y64).max(1)[fin]; ey_pwl = np.abs(out[2][0] - y64).max(1)[fin]
  nan_mismatch = int((np.isfinite(out[0][1]) != np.isfinite(out[2][1])).sum())
  # the table path must not be worse than the MLP kernel by more than a small factor / floor
  q = lambda e: (np.median(e), np.quantile(e, 0.999), e.max())
  fin_d = np.isfinite(lpd64) & np.isfinite(lpd[0]) & np.isfinite(lpd[2])
  ed_mlp = np.abs(lpd[0] - lpd64)[fin_d]; ed_pwl = np.abs(lpd[2] - lpd64)[fin_d]
  term_rels = [float(np.nanmax(np.abs(a - b) / np.maximum(np.abs(a), 1e-3))) for a, b in zip(terms[0], terms[2])]
  term_rel = max(term_rels)
  ok_extra = (np.quantile(ed_pwl, 0.999) <= 3 * np.quantile(ed_mlp, 0.999) + 5e-5 and np.median(ed_pwl) <= 3 * np.median(ed_mlp) + 1e-6)
  ok = ok_extra and (np.quantile(e_pwl, 0.999) <= 3 * np.quantile(e_mlp, 0.999) + 2e-5 and np.median(e_pwl) <= 3 * np.median(e_mlp) + 1e-6
        and np.quantile(ey_pwl, 0.999) <= 3 * np.quantile(ey_mlp, 0.999) + 2e-5 and nan_mismatch == 0)
  bad += 0 if ok else 1
  print(f"case {case:3d} kind {kind} scale {scale:4.2f} spread {spread:3.0f}: logp err mlp med/p999/max {q(e_mlp)[0]:.1e}/{q(e_mlp)[1]:.1e}/{q(e_mlp)[2]:.1e}"
        f"  tables {q(e_pwl)[0]:.1e}/{q(e_pwl)[1]:.1e}/{q(e_pwl)[2]:.1e}  y p999 {np.quantile(ey_mlp, .999):.1e}/{np.quantile(ey_pwl, .999):.1e}"
        f"  log_prob(data->base) p999 {np.quantile(ed_mlp, .999):.1e}/{np.quantile(ed_pwl, .999):.1e}  loss terms rel diff {term_rel:.1e} (kinetic {term_rels[0]:.1e} score {term_rels[1]:.1e} rkl {term_rels[2]:.1e}; sums {terms[0][0][0]:.3g} {terms[0][1][0]:.3g} {terms[0][2][0]:.3g})"
        f"  nonfinite mismatch {nan_mismatch} {'OK' if ok else 'WORSE'}")
print("cases:", n_cases, "table path worse than the MLP kernel in:", bad)
