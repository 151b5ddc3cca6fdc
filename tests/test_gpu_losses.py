"""GPU parity of the fused Monte-Carlo loss kernels (cnf_loss_terms, through
cnf_ot_amd.applications / utils) against oracle/losses.py, the float64
restatement of cnf_ot/mfc/applications.py and cnf_ot/utils.py:311-389, on the
SAME base noise (drawn by the HIP Philox kernel, copied to the host).

Tolerance: relative 2e-4 on each loss value.  The finite-difference terms
divide fp32 differences by dt = dx = 0.01: per-sample velocities/scores carry
~1e-5 / ~1e-3 of rounding noise, which averages out in the mean square but not
to fp32 epsilon.  Measured errors are printed.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RTOL = 2e-4


@pytest.fixture(scope="module")
def dev():
  assert torch.cuda.is_available()
  return torch.device("cuda", 0)


def _setup(dev, D=2, scale=0.2, seed=3):
  import oracle
  from oracle import losses as ol
  from cnf_ot_amd import FlowConfig, FlowModel, Params
  cfg = FlowConfig(dim=D)
  model = FlowModel(cfg)
  params = Params.random(cfg, scale, seed=seed, device=dev)
  flow = ol.OracleFlow(oracle.OracleConfig(D=D), params.flat.cpu().double().numpy())
  return model, params, flow


def _noise(model, params, seed, n):
  return model.terms_backend(params).normal(seed, n).cpu().double().numpy()


def _check(name, got, want, rtol=RTOL):
  got = float(got)
  rel = abs(got - want) / max(abs(want), 1e-12)
  print(f"[{name}] gpu={got:.8g} oracle={want:.8g} rel={rel:.2e}")
  assert rel <= rtol, (name, got, want, rel)


@pytest.mark.parametrize("spl", [1, 2, "tables"])
def test_single_terms_dim2(dev, spl):
  """spl = 1 / 2: the MLP loss kernel with one / two samples per lane;
  "tables": loss_pwl_kernel (conditioner from the piecewise-linear tables)."""
  from oracle import losses as ol
  from cnf_ot_amd import applications as app
  model, params, flow = _setup(dev)
  model.engine(dev).set_pwl(2 if spl == "tables" else 0)
  model.engine(dev).set_samples_per_lane(0 if spl == "tables" else spl)
  B, seed, t = 4096 + 37, 42, 0.4          # ragged batch: partial tiles
  z = _noise(model, params, seed, B)
  _check("kinetic", app.kinetic_loss_fn(model, 2, 0.01, params, t, seed, B), ol.kinetic_loss_fn(flow, 2, 0.01, t, z))
  _check("kinetic_score", app.kinetic_with_score_loss_fn(model, 2, 2.0, 0.01, 0.01, params, t, seed, B),
         ol.kinetic_with_score_loss_fn(flow, 2, 2.0, 0.01, 0.01, t, z))
  for sub in ("ou", "gradient", "nongradient"):
    _check(f"flow_matching[{sub}]", app.flow_matching_loss_fn(model, 2, 1.0, 0.5, sub, 0.01, 0.01, params, t, seed, B),
           ol.flow_matching_loss_fn(flow, 2, 1.0, 0.5, sub, t, z))
  for sub, a in (("quadratic", 0.0), ("double_well", 1.0), ("double_well", 0.5), ("obstacle", 0.0)):
    _check(f"potential[{sub}]", app.potential_loss_fn(model, 2, a, sub, params, t, seed, B),
           ol.potential_loss_fn(flow, a, sub, t, z), rtol=2e-5)
  for c in (0.0, 0.3, 1.0):
    _check(f"reverse_kl[c={c}]", app.reverse_kl_loss_fn(model, 2, 1.0, 4.0, params, c, seed, B),
           ol.reverse_kl_loss_fn(flow, 1.0, 4.0, c, z), rtol=5e-5)
  comp = app.draw_components(seed, B)
  for src in ("mixture", "gaussian"):
    for c in (0.0, 1.0):
      _check(f"kl[{src},c={c}]", app.kl_loss_fn(model, 2, 1.0, params, c, seed, B, source=src),
             ol.kl_loss_fn(flow, 1.0, c, z, src, comp), rtol=2e-5)
  model.engine(dev).set_samples_per_lane(0)
  model.engine(dev).set_pwl(1)


def test_composite_losses_reference_configs(dev):
  """BASELINE configs 1-3 in shape: OT free / obstacle (batch 4096), RWPO
  quadratic T=1 beta=1 (batch 131072 -> 4096 per slice)."""
  from oracle import losses as ol
  from cnf_ot_amd import applications as app
  model, params, flow = _setup(dev)
  seed = 42
  B = 4096
  z = _noise(model, params, seed, B)
  tb = app.draw_t_batch(seed, 4).astype(np.float64)
  comp = app.draw_components(seed, B)
  _check("ot_free", app.ot_loss_fn(model, 2, 1.0, 0.01, 4, "free", params, seed, 5000.0, B),
         ol.ot_loss_fn(flow, 2, 1.0, 0.01, "free", 5000.0, B, z, tb, "mixture", comp), rtol=2e-5)
  _check("ot_free_gaussian", app.ot_loss_fn(model, 2, 1.0, 0.01, 4, "free", params, seed, 5000.0, B, source="gaussian"),
         ol.ot_loss_fn(flow, 2, 1.0, 0.01, "free", 5000.0, B, z, tb, "gaussian"), rtol=2e-5)
  _check("ot_obstacle", app.ot_loss_fn(model, 2, 1.0, 0.01, 4, "obstacle", params, seed, 5000.0, B),
         ol.ot_loss_fn(flow, 2, 1.0, 0.01, "obstacle", 5000.0, B, z, tb, "mixture", comp), rtol=2e-5)
  B = 131072
  z = _noise(model, params, seed, B)
  tb = app.draw_t_batch(seed, 2, 1.0).astype(np.float64)
  _check("rwpo_quadratic", app.rwpo_loss_fn(model, 2, 1.0, 1.0, 0.01, 0.01, 2, "quadratic", 1.0, params, seed, 5000.0, B),
         ol.rwpo_loss_fn(flow, 2, 1.0, 1.0, 0.01, 0.01, "quadratic", 1.0, 5000.0, B, z, tb))
  B = 8192
  z = _noise(model, params, seed, B)
  tb = app.draw_t_batch(seed, 2, 2.0).astype(np.float64)
  _check("rwpo_double_well", app.rwpo_loss_fn(model, 2, 2.0, 10.0, 0.01, 0.01, 2, "double_well", 1.0, params, seed, 5000.0, B),
         ol.rwpo_loss_fn(flow, 2, 2.0, 10.0, 0.01, 0.01, "double_well", 1.0, 5000.0, B, z, tb))


def test_fokker_planck_dim10_and_lorenz_dim3(dev):
  """BASELINE config 4 in shape (D=10, OU drift a=1 sigma=0.5; a 32 768 shard),
  plus the 3-D Lorenz field."""
  from oracle import losses as ol
  from cnf_ot_amd import applications as app
  model, params, flow = _setup(dev, D=10, scale=0.12, seed=4)
  seed, B = 9, 32768
  z = _noise(model, params, seed, B)
  tb = app.draw_t_batch(seed, 2, 1.0).astype(np.float64)
  _check("fp_ou_d10", app.fp_loss_fn(model, 10, 1.0, 1.0, 0.5, 0.01, 0.01, 2, "ou", params, seed, 5000.0, B),
         ol.fp_loss_fn(flow, 10, 1.0, 1.0, 0.5, "ou", 5000.0, B, z, tb))
  with pytest.raises(ValueError):    # the reference's 'gradient' field is 2-D (applications.py:353-357)
    app.fp_loss_fn(model, 10, 1.0, 1.0, 0.5, 0.01, 0.01, 2, "gradient", params, seed, 5000.0, B)
  model3, params3, flow3 = _setup(dev, D=3, scale=0.15, seed=5)
  z3 = _noise(model3, params3, seed, 2048)
  _check("flow_matching[lorenz]", app.flow_matching_loss_fn(model3, 3, 1.0, 0.5, "lorenz", 0.01, 0.01, params3, 0.3, seed, 2048),
         ol.flow_matching_loss_fn(flow3, 3, 1.0, 0.5, "lorenz", 0.3, z3))


@pytest.mark.parametrize("pwl", [0, 2], ids=["mlp", "tables"])
def test_evaluators_fused_slices(dev, pwl):
  """utils.calc_kinetic_energy / calc_score_kinetic_energy: many slices per
  launch, each slice its own block of the noise stream."""
  from oracle import losses as ol
  from cnf_ot_amd import utils as amd_utils
  model, params, flow = _setup(dev)
  be = model.terms_backend(params)
  be.set_pwl(pwl)
  Bs, S = 8192, 7
  draw = lambda k: be.normal(5, Bs, first_sample=k * Bs).cpu().double().numpy()
  got = amd_utils.calc_kinetic_energy(model.apply.sample, params, 5, batch_size=Bs, t_size=S, dim=2,
                                      slices_per_launch=3)
  _check("calc_kinetic_energy", got, ol.calc_kinetic_energy(flow, 2, np.linspace(0, 1, S), draw))
  got = amd_utils.calc_score_kinetic_energy(model.apply.sample, model.apply.log_prob, params, 1.0, 2.0, 2, 5,
                                            batch_size=Bs, t_size=S, slices_per_launch=4)
  _check("calc_score_kinetic_energy", got, ol.calc_score_kinetic_energy(flow, 2, 2.0, np.linspace(0, 1, S), draw))


def test_identity_flow_closed_forms_at_full_size(dev):
  """Size-independent checks at BASELINE sizes: the identity flow has zero
  velocity, score -r, and reverse KL 0 against N(0, I)."""
  from cnf_ot_amd import FlowConfig, FlowModel, Params
  from cnf_ot_amd import applications as app
  cfg = FlowConfig(dim=2)
  model, params = FlowModel(cfg), Params.zeros(cfg, dev)
  B = 1 << 20
  assert float(app.kinetic_loss_fn(model, 2, 0.01, params, 0.5, 1, B)) == 0.0
  z = model.terms_backend(params).normal(1, B)
  want = float(((z.double() / 2.0) ** 2).mean())            # v = score/beta = -r/2; mean(v^2)*D/2, D=2
  got = float(app.kinetic_with_score_loss_fn(model, 2, 2.0, 0.01, 0.01, params, 0.5, 1, B))
  assert abs(got - want) / want <= 2e-3     # fp32 central difference of log_prob at dx = 0.01
  assert abs(float(app.reverse_kl_loss_fn(model, 2, 1.0, 4.0, params, 0.0, 1, B))) <= 1e-6
  got = float(app.potential_loss_fn(model, 2, 0.0, "quadratic", params, 1.0, 1, B))
  assert abs(got - float((z.double() ** 2).sum(1).mean() / 2)) <= 1e-6


def test_seeded_terms_equal_tensor_terms(dev):
  """cnf_loss_terms_seeded (noise drawn in the kernel) == cnf_loss_terms on the
  cnf_fill_normal tensor of the same stream positions."""
  from cnf_ot_amd import applications as app, _capi
  for D, scale in ((2, 0.2), (3, 0.15)):
    model, params, _ = _setup(dev, D=D, scale=scale)
    be = model.terms_backend(params)
    B, S, first = 1000, 5, 777          # ragged: partial tiles, unaligned Philox blocks
    ts = np.linspace(0.1, 0.9, S).astype(np.float32)
    for spec in (app._spec(_capi.TERM_KINETIC, dt=0.01), app._spec(_capi.TERM_POTENTIAL, subtype=2),
                 app._spec(_capi.TERM_REVERSE_KL, T=1.0, beta=4.0)):
      for spl in (1, 2, "tables"):
        if spl == "tables" and D != 2:
          continue
        be.set_pwl(2 if spl == "tables" else 0)
        be.set_samples_per_lane(0 if spl == "tables" else spl)
        own = torch.cat([be.normal(9, B, first_sample=first + s * B) for s in range(S)])
        a = be.loss_terms(spec, own, ts, B, False)
        b = be.loss_terms_seeded(spec, 9, ts, B, first_sample=first, slice_stride=B)
        assert torch.allclose(a, b, rtol=1e-12, atol=1e-9), (D, spec.kind, spl, a, b)
        shared = be.normal(9, B, first_sample=first)
        a = be.loss_terms(spec, shared, ts, B, True)
        b = be.loss_terms_seeded(spec, 9, ts, B, first_sample=first, slice_stride=0)
        assert torch.allclose(a, b, rtol=1e-12, atol=1e-9)
    be.set_samples_per_lane(0)
    be.set_pwl(1)


def test_config5_per_gpu_shape_properties(dev):
  """BASELINE configs[4] at its per-GPU shape: OT obstacle, dim 2, 131 072
  samples x 32 time-slices (kinetic + obstacle potential on the full slice; the
  8-GPU config is 2^20 x 32).  Too large for the float64 oracle in seconds, so
  size-independent properties: finite sums; noise drawn in the kernel == the
  same stream as a tensor; the table kernels == the MLP kernels (potential to
  1e-5 relative; the kinetic term is a sum of squared differences amplified by
  1/dt = 100, two fp32 evaluations of it agree to 5e-5); a 2-way sample
  sharding of every slice adds up to the whole."""
  from cnf_ot_amd import applications as app, _capi
  model, params, _ = _setup(dev)
  be = model.terms_backend(params)
  Bs, S = 131072, 32
  ts = np.linspace(0.0, 1.0, S).astype(np.float32)
  noise = be.normal(42, Bs)
  res = {}
  for pwl in (0, 2):
    be.set_pwl(pwl)
    for name, spec in (("kinetic", app._spec(_capi.TERM_KINETIC, dt=0.01)),
                       ("obstacle", app._spec(_capi.TERM_POTENTIAL, subtype=_capi.POTENTIALS["obstacle"]))):
      a = be.loss_terms(spec, noise, ts, Bs, True)
      assert be.last_path() == ("loss_tables" if pwl else "loss_mlp")
      b = be.loss_terms_seeded(spec, 42, ts, Bs, first_sample=0, slice_stride=0)
      assert torch.isfinite(a).all() and (a >= 0).all()
      assert torch.allclose(a, b, rtol=1e-12, atol=1e-9), (name, pwl)
      h = Bs // 2        # two ranks' blocks of every slice (distributed.shard_range)
      parts = be.loss_terms(spec, noise[:h].contiguous(), ts, h, True) + \
        be.loss_terms(spec, noise[h:].contiguous(), ts, h, True)
      assert torch.allclose(parts, a, rtol=1e-9), (name, pwl)
      res[(name, pwl)] = a
  for name in ("kinetic", "obstacle"):
    rel = ((res[(name, 0)] - res[(name, 2)]).abs() / res[(name, 0)].abs().clamp_min(1e-30)).max().item()
    print(f"\n[cfg5 {name}] tables vs mlp: max relative difference over {S} slices {rel:.2e}")
    assert rel <= (5e-5 if name == "kinetic" else 1e-5)
  be.set_pwl(1)
