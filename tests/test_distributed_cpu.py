"""The N > 1 path on CPU: two gloo ranks, the oracle standing in for the HIP
engine (tests/oracle_backend.py).  Checks that sample sharding + the single
sum all-reduce of the partial sums reproduce the one-rank loss, that the base
noise a rank draws is its block of the global stream, and that the host-side
loss composition equals the float64 restatement of applications.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
  if p not in sys.path:
    sys.path.insert(0, p)

from cnf_ot_amd import applications as app
from cnf_ot_amd import utils as amd_utils
from cnf_ot_amd.distributed import Shard, shard_range
from cnf_ot_amd.params import FlowConfig, Params


def _free_port():
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


def test_shard_range_tiles_the_batch():
  for n in (0, 1, 7, 64, 65536, 65537):
    for world in (1, 2, 3, 8):
      blocks = [shard_range(n, Shard(r, world)) for r in range(world)]
      assert blocks[0][0] == 0 and sum(c for _, c in blocks) == n
      for (s0, c0), (s1, _) in zip(blocks, blocks[1:]):
        assert s0 + c0 == s1
      assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1


def _losses(model, params, shard):
  dim, B = 2, 2048
  out = {}
  out["ot_free"] = app.ot_loss_fn(model, dim, 1.0, 0.01, 3, "free", params, 42, 5000.0, B, shard=shard)
  out["ot_obstacle"] = app.ot_loss_fn(model, dim, 1.0, 0.01, 2, "obstacle", params, 42, 5000.0, B,
                                      source="gaussian", shard=shard)
  out["rwpo"] = app.rwpo_loss_fn(model, dim, 1.0, 1.0, 0.01, 0.01, 2, "quadratic", 1.0, params, 7, 5000.0, B,
                                 shard=shard)
  out["fp"] = app.fp_loss_fn(model, dim, 1.0, 1.0, 0.5, 0.01, 0.01, 2, "ou", params, 9, 5000.0, B, shard=shard)
  out["kin_energy"] = amd_utils.calc_kinetic_energy(model, params, 3, batch_size=512, t_size=5, dim=dim,
                                                    slices_per_launch=2, shard=shard)
  return {k: float(v) for k, v in out.items()}


def _worker(rank, world, port, q):
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    from oracle_backend import OracleModel
    cfg = FlowConfig(dim=2)
    params = Params.random(cfg, 0.2, seed=5)
    res = _losses(OracleModel(cfg), params, None)       # shard from the process group
    q.put((rank, res))
  finally:
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_rank(oracle_lib):
  from oracle_backend import OracleModel
  cfg = FlowConfig(dim=2)
  params = Params.random(cfg, 0.2, seed=5)
  single = _losses(OracleModel(cfg), params, Shard())
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  results = dict(q.get(timeout=300) for _ in range(2))
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  for rank in (0, 1):
    for k, v in single.items():
      assert abs(results[rank][k] - v) <= 1e-9 * max(1.0, abs(v)), (rank, k, results[rank][k], v)


def _vg_worker(rank, world, port, q):
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    from oracle_backend import OracleModel
    cfg = FlowConfig(dim=2)
    params = Params.random(cfg, 0.2, seed=5)
    model = OracleModel(cfg)
    out = {}
    for name, overlap in (("blocking", False), ("overlapped", True)):
      g = torch.zeros(cfg.param_count(), dtype=torch.float64)
      loss = app.ot_loss_fn(model, 2, 1.0, 0.01, 2, "obstacle", params, 42, 5000.0, 256, source="gaussian", grad=g,
                            overlap=overlap)
      out[name] = (float(loss), g.numpy().copy())
    # the captured step is single-rank: under a process group it says so instead of recording a collective
    from cnf_ot_amd import solvers
    try:
      solvers.CapturedUpdate(lambda *a, **k: None, solvers.Adam(1e-3), 256)(params, 1, 1.0, None)
      out["captured"] = "ran"
    except NotImplementedError as e:
      out["captured"] = str(e)
    q.put((rank, out))
  finally:
    dist.destroy_process_group()


def test_overlapped_allreduce_equals_the_blocking_one(oracle_lib):
  """BASELINE.json configs[4] ("allreduce/compute overlap"): the all-reduce of the density-fit terms started
  asynchronously under the kinetic / obstacle slices + a second one for those (applications._Ctx.reduce_begin /
  reduce_end) against ONE blocking collective at the end -- two gloo ranks, loss AND gradient, and both against the
  single-rank evaluation."""
  from oracle_backend import OracleModel
  cfg = FlowConfig(dim=2)
  params = Params.random(cfg, 0.2, seed=5)
  g1 = torch.zeros(cfg.param_count(), dtype=torch.float64)
  l1 = float(app.ot_loss_fn(OracleModel(cfg), 2, 1.0, 0.01, 2, "obstacle", params, 42, 5000.0, 256, source="gaussian",
                            grad=g1, shard=Shard()))
  assert np.count_nonzero(g1.numpy()) > 20
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_vg_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  results = dict(q.get(timeout=600) for _ in range(2))
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  scale = np.abs(g1.numpy()).max()
  for rank in (0, 1):
    lb, gb = results[rank]["blocking"]
    lo, go = results[rank]["overlapped"]
    assert abs(lo - lb) <= 1e-12 * abs(lb) and np.abs(go - gb).max() <= 1e-12 * scale
    assert abs(lb - l1) <= 1e-9 * abs(l1) and np.abs(gb - g1.numpy()).max() <= 1e-6 * scale      # (finite-difference gradients)
    assert "single rank" in results[rank]["captured"]


def test_host_composition_matches_reference_restatement(oracle_lib):
  """applications.* over the oracle backend == oracle/losses.py (the float64
  restatement of cnf_ot/mfc/applications.py) given the same draws."""
  import oracle
  from oracle import losses as ol
  from oracle_backend import OracleBackend, OracleModel
  cfg = FlowConfig(dim=2)
  params = Params.random(cfg, 0.2, seed=11)
  model = OracleModel(cfg)
  be = OracleBackend(cfg, params.flat.double().numpy())
  flow = be.flow
  B, seed = 2048, 42
  z = be.normal(seed, B).double().numpy()
  tb = app.draw_t_batch(seed, 3).astype(np.float64)
  comp = app.draw_components(seed, B)
  got = float(app.ot_loss_fn(model, 2, 1.0, 0.01, 3, "free", params, seed, 5000.0, B, shard=Shard()))
  want = ol.ot_loss_fn(flow, 2, 1.0, 0.01, "free", 5000.0, B, z, tb, "mixture", comp)
  assert abs(got - want) <= 1e-9 * abs(want)
  got = float(app.ot_loss_fn(model, 2, 1.0, 0.01, 3, "obstacle", params, seed, 5000.0, B, source="gaussian",
                             shard=Shard()))
  want = ol.ot_loss_fn(flow, 2, 1.0, 0.01, "obstacle", 5000.0, B, z, tb, "gaussian")
  assert abs(got - want) <= 1e-7 * abs(want)     # the host mixes source/target samples in float32
  tb2 = app.draw_t_batch(seed, 2, 2.0).astype(np.float64)
  got = float(app.rwpo_loss_fn(model, 2, 2.0, 10.0, 0.01, 0.01, 2, "double_well", 1.0, params, seed, 5000.0, B,
                               shard=Shard()))
  want = ol.rwpo_loss_fn(flow, 2, 2.0, 10.0, 0.01, 0.01, "double_well", 1.0, 5000.0, B, z, tb2)
  assert abs(got - want) <= 1e-9 * abs(want)
  tb1 = app.draw_t_batch(seed, 2, 1.0).astype(np.float64)
  got = float(app.fp_loss_fn(model, 2, 1.0, 1.0, 0.5, 0.01, 0.01, 2, "gradient", params, seed, 5000.0, B,
                             shard=Shard()))
  want = ol.fp_loss_fn(flow, 2, 1.0, 1.0, 0.5, "gradient", 5000.0, B, z, tb1)
  assert abs(got - want) <= 1e-9 * abs(want)
  # evaluator: per-slice key split == consecutive blocks of the stream
  got = float(amd_utils.calc_kinetic_energy(model, params, 3, batch_size=256, t_size=4, dim=2, shard=Shard()))
  want = ol.calc_kinetic_energy(flow, 2, np.linspace(0, 1, 4),
                                lambda k: be.normal(3, 256, first_sample=k * 256).double().numpy())
  assert abs(got - want) <= 1e-7 * max(abs(want), 1e-12)   # slice times travel as float32


def test_loss_argument_errors(oracle_lib):
  from oracle_backend import OracleModel
  cfg = FlowConfig(dim=3)
  params = Params.zeros(cfg)
  model = OracleModel(cfg)
  with pytest.raises(Exception, match="nongradient"):      # applications.py:359-360
    app.fp_loss_fn(model, 3, 1.0, 1.0, 0.5, 0.01, 0.01, 1, "nongradient", params, 0, 1.0, 64, shard=Shard())
  with pytest.raises(ValueError):
    app.potential_loss_fn(model, 3, 1.0, "cubic", params, 0.5, 0, 64, shard=Shard())
  with pytest.raises(ValueError):                          # mixture source is 2-D
    app.kl_loss_fn(model, 3, 1.0, params, 0.0, 0, 64, shard=Shard())
