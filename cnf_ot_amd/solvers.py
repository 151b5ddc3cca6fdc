"""Mirror of the training step of cnf_ot/mfc/solvers.py (the caller of the hot
path): config keys of config/mfc.yaml, model construction (:41-56), the loss
binding (:58-88), and `update` = value_and_grad + Adam (:90-97), on the HIP
kernels.  Plotting / printing / post-training evaluation of the reference
(:129-493) are out of scope; the two evaluators are
in cnf_ot_amd.utils.
"""
from dataclasses import dataclass, field
from functools import partial
from typing import Any, Callable, Dict, Tuple

import torch

from . import _capi, applications
from .flows import DeviceRng, RQSFlow, FlowModel, _OnDevice, _stream_ptr, mark_updated
from .params import Params

# config/mfc.yaml:6-40 (the checked-in defaults)
DEFAULT_CONFIG: Dict[str, Dict[str, Any]] = {
  "general": {"type": "rwpo", "dim": 2, "dx": 0.01, "dt": 0.01, "t_batch_size": 1, "seed": 42},
  "ot": {"subtype": "free"},
  "rwpo": {"T": 2, "beta": 10, "a": 1, "pot_type": "double_well"},
  "fp": {"T": 1, "a": 1, "sigma": 0.5, "velocity_field_type": "gradient"},
  "cnf": {"flow_num_layers": 2, "mlp_num_layers": 2, "hidden_size": 16, "num_bins": 5},
  "train": {"epochs": 30000, "lr": 0.001, "_lambda": 5000.0, "batch_size": 2048, "eval_frequency": 100},
}


def load_config(path: str = None, overrides: Dict[str, Dict[str, Any]] = None) -> Dict[str, Dict[str, Any]]:
  """yaml.safe_load(config/mfc.yaml) (solvers.py:496-500) merged over the
  defaults; the unused `hydra:` block is ignored."""
  cfg = {k: dict(v) for k, v in DEFAULT_CONFIG.items()}
  if path is not None:
    import yaml
    with open(path) as f:
      loaded = yaml.safe_load(f) or {}
    for sec, vals in loaded.items():
      if sec in cfg and isinstance(vals, dict):
        cfg[sec].update(vals)
  for sec, vals in (overrides or {}).items():
    cfg.setdefault(sec, {}).update(vals)
  return cfg


def build_model(config) -> FlowModel:
  """solvers.py:41-48"""
  c = config["cnf"]
  return RQSFlow(event_shape=(config["general"]["dim"],), num_layers=c["flow_num_layers"],
                 hidden_sizes=[c["hidden_size"]] * c["mlp_num_layers"], num_bins=c["num_bins"], periodized=False)


def bind_loss(config, model) -> Callable:
  """solvers.py:58-88: loss_fn(params, rng, _lambda, batch_size)."""
  g = config["general"]
  _type, dim, dt, dx, tbs = g["type"], g["dim"], g["dt"], g["dx"], g["t_batch_size"]
  if _type == "rwpo":
    r = config["rwpo"]
    return partial(applications.rwpo_loss_fn, model, dim, r["T"], r["beta"], dt, dx, tbs, r["pot_type"], r["a"])
  if _type == "fp":
    f = config["fp"]
    return partial(applications.fp_loss_fn, model, dim, f["T"], f["a"], f["sigma"], dt, dx, tbs,
                   f["velocity_field_type"])
  if _type == "ot":
    return partial(applications.ot_loss_fn, model, dim, 1, dt, tbs, config["ot"]["subtype"])
  raise Exception(f"Unknown problem type: {_type}...")        # solvers.py:87-88


@dataclass
class AdamState:
  """optax.adam state: step count and the two moment estimates (flat tensors)."""
  step: int
  mu: torch.Tensor
  nu: torch.Tensor


@dataclass
class Adam:
  """optax.adam(lr) (solvers.py:55): b1 = 0.9, b2 = 0.999, eps = 1e-8."""
  lr: float
  b1: float = 0.9
  b2: float = 0.999
  eps: float = 1e-8

  def init(self, params: Params) -> AdamState:
    return AdamState(0, torch.zeros_like(params.flat), torch.zeros_like(params.flat))

  def apply(self, params: Params, grads: Params, state: AdamState, step_state: torch.Tensor = None) -> AdamState:
    """optimizer.update + optax.apply_updates (solvers.py:95-96), in place on
    params.flat (one kernel: cnf_adam_step).  step_state: a DeviceRng's `state` -- the step count is then read on the
    device (cnf_adam_step_dev; cnf_step_begin has already counted this step), as a captured step needs it."""
    lib = _capi.lib()
    state.step += 1
    dev = params.flat.device
    with _OnDevice(dev):
      if step_state is not None:
        _capi.check(lib.cnf_adam_step_dev(params.flat.data_ptr(), grads.flat.data_ptr(), state.mu.data_ptr(),
                                          state.nu.data_ptr(), params.flat.numel(), self.lr, self.b1, self.b2, self.eps,
                                          step_state.data_ptr(), _stream_ptr(dev)), "cnf_adam_step_dev")
      else:
        _capi.check(lib.cnf_adam_step(params.flat.data_ptr(), grads.flat.data_ptr(), state.mu.data_ptr(),
                                      state.nu.data_ptr(), params.flat.numel(), self.lr, self.b1, self.b2, self.eps,
                                      state.step, _stream_ptr(dev)), "cnf_adam_step")
    # the kernel wrote params.flat behind torch's back: engines must re-prepare (FlowEngine.load)
    mark_updated(params.flat)
    return state


class CapturedUpdate:
  """`update` of solvers.py:90-97 as ONE device-side program -- what the reference's jitted step is: value_and_grad
  + Adam captured into a HIP graph on the first calls and replayed afterwards (a default-config step is ~20 small
  launches: launch latency, not kernels).  Everything that changes from step to step is read from device memory:
  the step's key (one 8-byte copy per call), the time batch, the mixture components and the base noise drawn from it
  by kernels, Adam's step count (include/cnf_ot_amd.h, "a training step as one device-side program").

  Same call surface as the eager update: (params, rng, _lambda, opt_state) -> (loss, params, opt_state), with
  `params` / `opt_state` updated in place and `loss` a 0-dim device tensor that the NEXT call overwrites.  The
  graph holds the addresses of `params.flat` and of the optimiser state and the value of `_lambda`: call it with the
  same objects (checked).  `replay=False` runs the very same body eagerly (what the graph is compared with)."""

  WARMUP = 2       # eager steps before the capture: allocations (gradient slabs, cached weight vectors) happen there

  def __init__(self, loss_fn: Callable, optimizer: Adam, batch_size: int, replay: bool = True):
    self.vg = applications.value_and_grad(loss_fn)
    self.opt, self.B, self.replay = optimizer, batch_size, replay
    self.rng, self.graph, self.loss, self._key, self._calls = None, None, None, None, 0
    # The warm-up steps run on the stream the graph is captured on: the engine's table workspaces are reserved PER
    # STREAM and cannot grow during a capture (FlowEngine.reserve), so a step warmed up on the caller's stream and
    # captured on torch's side stream found no tables there and recorded the MLP kernels -- correct, and for a
    # large batch several times slower than the eager step (found by the ot_large case of
    # test_captured_step_equals_the_eager_step_bit_for_bit).
    self._stream = None

  def _body(self, params, _lambda, opt_state):
    dev = params.flat.device
    with _OnDevice(dev):
      _capi.check(_capi.lib().cnf_step_begin(self.rng.ptr, _stream_ptr(dev)), "cnf_step_begin")
    loss, grads = self.vg(params, self.rng, _lambda, self.B)
    self.opt.apply(params, grads, opt_state, step_state=self.rng.state)
    return loss

  def __call__(self, params: Params, rng, _lambda, opt_state: AdamState):
    if self.rng is None:
      from .distributed import current_shard
      if self.replay and current_shard().world > 1:
        # (the loss's all-reduce would be recorded into the graph: RCCL can, gloo cannot, and neither has been run here)
        raise NotImplementedError("CapturedUpdate: single rank only -- use the eager update under torch.distributed")
      self.rng = DeviceRng(params.flat.device)
    key = (params.flat.data_ptr(), opt_state.mu.data_ptr(), opt_state.nu.data_ptr(), float(_lambda))
    if self._key is not None and key != self._key:
      raise ValueError("CapturedUpdate was captured for other parameter / optimiser tensors or another _lambda")
    self._key = key
    self.rng.set_key(rng)
    self._calls += 1
    if not self.replay:
      self.loss = self._body(params, _lambda, opt_state)
    elif self._calls <= self.WARMUP:
      dev = params.flat.device
      if self._stream is None:
        self._stream = torch.cuda.Stream(device=dev)
      cur = torch.cuda.current_stream(dev)
      self._stream.wait_stream(cur)
      with torch.cuda.stream(self._stream):
        self.loss = self._body(params, _lambda, opt_state)
      cur.wait_stream(self._stream)
    elif self.graph is None:
      torch.cuda.synchronize(params.flat.device)
      graph = torch.cuda.CUDAGraph()
      step0 = opt_state.step
      with torch.cuda.graph(graph, stream=self._stream):
        self.loss = self._body(params, _lambda, opt_state)
      opt_state.step = step0            # (the capture only recorded the step)
      self.graph = graph
      graph.replay()
      opt_state.step += 1
    else:
      self.graph.replay()
      opt_state.step += 1
    return self.loss, params, opt_state


def make_update(loss_fn: Callable, optimizer: Adam, batch_size: int, capture: bool = False) -> Callable:
  """`update` of solvers.py:90-97.  The reference returns new pytrees; here the
  parameters and the optimiser state are updated IN PLACE (the same objects are
  returned), which is what `params, opt_state = update(...)` callers expect.
  capture=True: the step as one replayed HIP graph (CapturedUpdate)."""
  if capture:
    return CapturedUpdate(loss_fn, optimizer, batch_size)
  vg = applications.value_and_grad(loss_fn)

  def update(params: Params, rng, _lambda, opt_state: AdamState) -> Tuple[torch.Tensor, Params, AdamState]:
    loss, grads = vg(params, rng, _lambda, batch_size)
    opt_state = optimizer.apply(params, grads, opt_state)
    return loss, params, opt_state

  return update


def train(config, epochs: int = None, log=None, capture: bool = False):
  """The training loop of solvers.py:99-127 without tqdm/plots: returns
  (params, loss history as a list of 0-dim device tensors).  capture=True: every step after the first two is the
  replay of one HIP graph (CapturedUpdate)."""
  model = build_model(config)
  seed = int(config["general"]["seed"])
  params = model.init(seed)
  opt = Adam(config["train"]["lr"])
  state = opt.init(params)
  update = make_update(bind_loss(config, model), opt, config["train"]["batch_size"], capture=capture)
  n = config["train"]["epochs"] if epochs is None else epochs
  hist = []
  for step in range(n):
    # update_rng, rng = jax.random.split(rng) (solvers.py:104): an independent Philox key per step
    step_rng = (seed + 0x9E3779B97F4A7C15 * (step + 1)) & 0xFFFFFFFFFFFFFFFF
    loss, params, state = update(params, step_rng, config["train"]["_lambda"], state)
    hist.append(loss.clone() if capture else loss)       # (a captured step overwrites its loss tensor)
    if log is not None and step % config["train"]["eval_frequency"] == 0:
      log(step, float(loss))
  return model, params, hist
