"""Mirror of the two Monte-Carlo evaluators of cnf_ot/utils.py (:311-389), the
callers that define BASELINE's benchmark shape: 10 000 time-slices x 65 536
samples.  The reference issues 2-3 jitted sample calls (+ 2*dim log_prob calls)
per slice from a Python loop; here a chunk of slices is ONE fused launch
(grid over (slice, sample tile)) that returns one sum per slice.
"""
from typing import Optional

import numpy as np
import torch

from . import _capi
from .applications import _spec
from .distributed import Shard, all_reduce_sums, current_shard, shard_range


def _model_of(fn_or_model):
  if hasattr(fn_or_model, "terms_backend"):
    return fn_or_model
  owner = getattr(fn_or_model, "__self__", None)      # model.apply.sample is a bound method
  if owner is not None and hasattr(owner, "_m"):
    return owner._m
  raise TypeError("expected a cnf_ot_amd FlowModel or one of its model.apply functions")


def _mc_energy(model, params, rng, spec, batch_size, t_array, dim, slices_per_launch, shard):
  shard = shard if shard is not None else current_shard()
  be = model.terms_backend(params)
  start, count = shard_range(batch_size, shard)
  total = torch.zeros(1, dtype=torch.float64, device=be.device)
  t_array = np.asarray(t_array, dtype=np.float32)
  for k0 in range(0, len(t_array), slices_per_launch):
    ts = t_array[k0:k0 + slices_per_launch]
    # slice k draws its own noise (the key split of utils.py:328): samples
    # [k*batch_size, (k+1)*batch_size) of the seed's stream; this rank's block of each
    if hasattr(be, "loss_terms_seeded"):      # HIP engine: noise is drawn inside the kernel, never in HBM
      total += be.loss_terms_seeded(spec, rng, ts, count, first_sample=k0 * batch_size + start,
                                    slice_stride=batch_size).sum()
      continue
    noise = torch.empty(len(ts) * count, dim, dtype=torch.float32, device=be.device)
    for j in range(len(ts)):
      noise[j * count:(j + 1) * count] = be.normal(rng, count, first_sample=(k0 + j) * batch_size + start)
    total += be.loss_terms(spec, noise, ts, count, False).sum()
  total = all_reduce_sums(total, shard)
  # e_kin += mean(velocity**2) / 2 per slice; return e_kin / t_size * dim   (utils.py:338-340)
  return total[0] / (batch_size * dim) / 2 / len(t_array) * dim


def calc_kinetic_energy(sample_fn, params, rng, batch_size: int = 65536, t_size: int = 10000, dim: int = 1,
                        slices_per_launch: int = 1024, shard: Optional[Shard] = None):
  """cnf_ot/utils.py:311-340 (dt = 0.01 hard-coded there, :324)."""
  model = _model_of(sample_fn)
  t_array = np.linspace(0.0, 1.0, t_size)
  return _mc_energy(model, params, rng, _spec(_capi.TERM_KINETIC, dt=0.01), batch_size, t_array, dim,
                    slices_per_launch, shard)


def calc_score_kinetic_energy(sample_fn, log_prob_fn, params, T: float = 1, beta: float = 1, dim: int = 1,
                              rng=0, batch_size: int = 65536, t_size: int = 10000,
                              slices_per_launch: int = 1024, shard: Optional[Shard] = None):
  """cnf_ot/utils.py:343-389 (dt = dx = 0.01 hard-coded, :360,378)."""
  model = _model_of(sample_fn)
  t_array = np.linspace(0.0, T, t_size)
  spec = _spec(_capi.TERM_KINETIC_SCORE, dt=0.01, dx=0.01, coef=1.0 / beta)
  return _mc_energy(model, params, rng, spec, batch_size, t_array, dim, slices_per_launch, shard)
