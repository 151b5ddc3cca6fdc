"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/cnf_ot_amd.h declares (no compute calls), the host-side
parameter container follows the reference's haiku tree, and the host mirror
validates arguments like the reference does."""
import os
import re

import numpy as np
import pytest
import torch

import cnf_ot_amd
from cnf_ot_amd import _capi
from cnf_ot_amd.params import FlowConfig, Params, param_spec, from_tree, flatten

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
  with open(os.path.join(ROOT, "include", header)) as f:
    text = f.read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return set(re.findall(r"\b(cnf_[a-z_0-9]+)\s*\(", text))


def test_library_exports_every_declared_symbol():
  """Both headers: the drop-in boundary (cnf_ot_amd.h <-> _capi.SYMBOLS) and the test / measurement knobs
  (cnf_ot_amd_debug.h <-> _capi._INTERNAL) -- nothing is bound that no header declares."""
  lib = _capi.lib()
  for header, table in (("cnf_ot_amd.h", _capi.SYMBOLS), ("cnf_ot_amd_debug.h", _capi._INTERNAL)):
    declared = _declared(header)
    assert declared, header
    assert declared == set(table), (header, declared ^ set(table))
    for name in declared:
      assert hasattr(lib, name), name


def test_c_param_count_and_support_table():
  lib = _capi.lib()
  cfg = _capi.CnfConfig()
  lib.cnf_config_default(cfg, 2)
  assert (cfg.num_layers, cfg.hidden_size, cfg.mlp_num_layers, cfg.num_bins) == (2, 16, 2, 5)
  assert abs(cfg.range_min + 10) < 1e-9 and abs(cfg.range_max - 10) < 1e-9
  assert lib.cnf_param_count(cfg) == 1200
  lib.cnf_config_default(cfg, 10)
  assert lib.cnf_param_count(cfg) == 11824
  assert lib.cnf_config_supported(cfg) == 1
  cfg.hidden_size = 17
  assert lib.cnf_config_supported(cfg) == 0
  cfg.hidden_size = 16
  cfg.num_bins = 0
  assert lib.cnf_param_count(cfg) == _capi.CNF_ERR_INVALID
  assert lib.cnf_build_arch() == b"gfx950"
  assert b"invalid" in lib.cnf_strerror(_capi.CNF_ERR_INVALID)


def test_param_tree_follows_haiku_names():
  cfg = FlowConfig(dim=3)
  names = [(m, n) for m, n, _ in param_spec(cfg)]
  assert names[0] == ("~", "first")
  assert ("mlp_layer0_d1/~/linear_0", "w") in names and ("mlp_layer1_d2/~/linear_1", "b") in names
  assert ("linear_out_layer1_d2", "w") in names
  shapes = {(m, n): s for m, n, s in param_spec(cfg)}
  assert shapes[("~", "first")] == (1, 16)
  assert shapes[("mlp_layer0_d2/~/linear_0", "w")] == (3, 16)   # [c, y0, y1] -> 16
  assert shapes[("linear_out_layer0_d1", "w")] == (16, 16)
  assert FlowConfig(dim=2).param_count() == 1200 and FlowConfig(dim=10).param_count() == 11824


def test_params_views_share_flat_storage_and_roundtrip(tmp_path):
  cfg = FlowConfig(dim=2)
  p = Params.random(cfg, 0.3, seed=1)
  p["~"]["first"][0, 3] = 7.0
  assert p.flat[3] == 7.0
  path = str(tmp_path / "p.npz")
  p.save_npz(path)
  q = Params.load_npz(cfg, path)
  assert torch.equal(p.flat, q.flat)
  nested = {m: {n: t.double().numpy() for n, t in d.items()} for m, d in p.items()}
  assert torch.equal(from_tree(cfg, nested).flat, p.flat)
  assert flatten(cfg, p, "cpu").data_ptr() == p.flat.data_ptr()       # zero-copy
  with pytest.raises(ValueError):
    from_tree(cfg, {**nested, "~": {"first": np.zeros((1, 15))}})


def test_init_is_identity_flow_parameters():
  p = Params.init(FlowConfig(dim=3), seed=0)
  assert torch.count_nonzero(p["~"]["first"]) == 0
  for mod, leaves in p.items():
    if mod.startswith("linear_out_"):
      assert torch.count_nonzero(leaves["w"]) == 0 and torch.count_nonzero(leaves["b"]) == 0
    if mod.startswith("mlp_"):
      assert torch.count_nonzero(leaves["w"]) > 0 and torch.count_nonzero(leaves["b"]) == 0


def test_rqsflow_signature_and_argument_errors():
  m = cnf_ot_amd.RQSFlow(event_shape=(2,), num_layers=2, hidden_sizes=[16] * 2, num_bins=5, periodized=False)
  assert m.cfg == FlowConfig(dim=2)
  assert cnf_ot_amd.Flow._fields == ("log_prob", "sample", "sample_and_log_prob", "forward", "inverse",
                                     "forward_jac", "inverse_jac", "gauge_potential")   # flows.py:215-219
  # periodized=True (flows.py:58-64,127-131): the torus model -- range [0, 2 pi], and a first linear layer with
  # 2 (1 + d) rows (sin and cos of every conditioner input)
  mp = cnf_ot_amd.RQSFlow((3,), 2, [16, 16], 5, periodized=True)
  assert mp.cfg.periodized and mp.cfg.range_min == 0.0 and abs(mp.cfg.range_max - 2 * np.pi) < 1e-12
  shapes = {(mod, name): shape for mod, name, shape in param_spec(mp.cfg)}
  assert shapes[("mlp_layer0_d1/~/linear_0", "w")] == (4, 16) and shapes[("mlp_layer1_d2/~/linear_0", "w")] == (6, 16)
  assert mp.cfg.param_count() == FlowConfig(dim=3).param_count() + 2 * (2 + 3) * 16
  with pytest.raises(NotImplementedError):
    cnf_ot_amd.RQSFlow((2,), 2, [16, 32], 5)


def test_no_cpu_fallback_in_product_path():
  """The product package must not import the oracle."""
  import subprocess, sys
  code = ("import sys; import cnf_ot_amd, cnf_ot_amd.flows, cnf_ot_amd.params; "
          "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'")
  subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)
  for dirpath, _, files in os.walk(os.path.join(ROOT, "cnf_ot_amd")):
    for fn in files:
      if fn.endswith((".py", ".hip", ".h", ".cpp")):
        with open(os.path.join(dirpath, fn)) as f:
          src = f.read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn


def test_bench_starts_its_own_ranks_as_child_processes(monkeypatch):
  """`python bench.py --gpus N` without a launcher: the N ranks are CHILD processes started through
  torch.distributed.run before any GPU call of the parent (never an exec of a process that touched the GPU)."""
  import subprocess, sys, types
  sys.path.insert(0, ROOT)
  import bench
  seen = {}

  def fake_run(cmd, env=None, **kw):
    seen["cmd"], seen["env"] = cmd, env
    return types.SimpleNamespace(returncode=7)

  monkeypatch.setattr(subprocess, "run", fake_run)
  monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "3"])
  monkeypatch.delenv("WORLD_SIZE", raising=False)
  with pytest.raises(SystemExit) as exc:
    bench.main()
  assert exc.value.code == 7                                   # the children's return code is handed back
  cmd = seen["cmd"]
  assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
  assert "--nproc-per-node=2" in cmd and "127.0.0.1" in cmd
  assert cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:] == sys.argv[1:]
  assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
  # without --share-device a box with fewer GPUs than ranks is refused before anything is started
  seen.clear()
  monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "64"])
  with pytest.raises(SystemExit) as exc:
    bench.main()
  assert exc.value.code == 2 and not seen
