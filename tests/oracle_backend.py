"""Test-only stand-ins that give the CPU oracle the interface of the HIP engine
(`terms_backend(params)` -> object with `.normal`, `.loss_terms`, `.device`), so
that the host logic of cnf_ot_amd.applications / utils / distributed -- loss
composition, sample sharding, the single all-reduce -- can be exercised on CPU
(gloo, world_size 2) without a GPU.  TEST INFRASTRUCTURE: never imported by the
product package."""
import numpy as np
import torch

import oracle
from oracle import losses as ol
from cnf_ot_amd import _capi
from cnf_ot_amd.flows import seed_to_u64
from cnf_ot_amd.params import FlowConfig, flatten

_POT = {v: k for k, v in _capi.POTENTIALS.items()}
_DRIFT = {0: "ou", 1: "gradient", 2: "nongradient", 3: "lorenz"}


class OracleBackend:
  def __init__(self, cfg: FlowConfig, flat_params):
    self.cfg = cfg
    self.device = torch.device("cpu")
    ocfg = oracle.OracleConfig(D=cfg.dim, L=cfg.num_layers, H=cfg.hidden_size, M=cfg.mlp_num_layers,
                               K=cfg.num_bins)
    self.flow = ol.OracleFlow(ocfg, np.asarray(flat_params, dtype=np.float64))

  def normal(self, seed, n_samples, first_sample=0):
    seed, off = seed_to_u64(seed)
    D = self.cfg.dim
    z = oracle.normal(seed, (off + first_sample) * D, n_samples * D).reshape(n_samples, D)
    return torch.from_numpy(z.astype(np.float32))      # what the HIP stream produces, up to rounding

  def loss_terms(self, spec, pts, t, B, shared):
    pts = np.asarray(pts.double().numpy() if torch.is_tensor(pts) else pts, dtype=np.float64)
    t = np.atleast_1d(np.asarray(t, dtype=np.float64))
    D, f = self.cfg.dim, self.flow
    out = np.zeros(len(t))
    for s, ts in enumerate(t):
      z = pts if shared else pts[s * B:(s + 1) * B]
      k = spec.kind
      if k == _capi.TERM_KINETIC:
        out[s] = ol.kinetic_loss_fn(f, D, spec.dt, ts, z) * 2 / D * (B * D)
      elif k == _capi.TERM_KINETIC_SCORE:
        out[s] = ol.kinetic_with_score_loss_fn(f, D, 1.0 / spec.coef, spec.dt, spec.dx, ts, z) * 2 / D * (B * D)
      elif k == _capi.TERM_FLOW_MATCHING:
        out[s] = ol.flow_matching_loss_fn(f, D, spec.a, spec.coef, _DRIFT[spec.subtype], ts, z) * 2 / D * (B * D)
      elif k == _capi.TERM_POTENTIAL:
        out[s] = ol.potential_loss_fn(f, spec.a, _POT[spec.subtype], ts, z) * B
      elif k == _capi.TERM_REVERSE_KL:
        out[s] = ol.reverse_kl_loss_fn(f, spec.T, spec.beta, ts, z) * B
      elif k == _capi.TERM_NEG_LOGPROB:
        out[s] = -f.log_prob(z, [ts]).sum()
      else:
        raise ValueError(k)
    return torch.from_numpy(out)


  # value_and_grad plumbing on the CPU: the gradient of the term's sum by central differences of the oracle, on a
  # fixed subset of the parameters (the rest stay zero) -- enough to check that sharded gradients are reduced,
  # phased (reduce_begin / reduce_end) and composed like single-rank ones
  GRAD_SUBSET = np.arange(0, 1200, 37)

  def loss_terms_grad(self, spec, pts, t, B, shared, scale, grad, sums=None):
    base = self.loss_terms(spec, pts, t, B, shared)
    p0 = self.flow.params.copy()
    h = 1e-6
    g = np.zeros(grad.numel())
    for i in self.GRAD_SUBSET[self.GRAD_SUBSET < grad.numel()]:
      vals = []
      for sgn in (1.0, -1.0):
        q = p0.copy(); q[i] += sgn * h
        self.flow = ol.OracleFlow(self.flow.cfg, q)
        vals.append(float(self.loss_terms(spec, pts, t, B, shared).sum()))
      g[i] = (vals[0] - vals[1]) / (2 * h)
    self.flow = ol.OracleFlow(self.flow.cfg, p0)
    grad.add_(torch.from_numpy(scale * g).to(grad.dtype))
    return base


class OracleModel:
  """Duck-types cnf_ot_amd.FlowModel for the loss / evaluator code."""

  def __init__(self, cfg: FlowConfig):
    self.cfg = cfg

  def terms_backend(self, params, device=None):
    return OracleBackend(self.cfg, flatten(self.cfg, params, "cpu").double().numpy())
