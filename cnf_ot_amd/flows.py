"""Host-side mirror of cnf_ot's flow-model call surface on PyTorch-ROCm tensors.

``RQSFlow(...)`` here returns what the reference gets from
``hk.without_apply_rng(hk.multi_transform(RQSFlow(...)))``
(cnf_ot/models/flows.py:178-226, cnf_ot/mfc/solvers.py:41-54): an object with
``init(rng, x, c) -> params`` and ``apply.<fn>(params, ...)`` for the eight
functions of the reference's ``Flow`` namedtuple.  All arithmetic runs in the
hand-written HIP kernels behind the C ABI (include/cnf_ot_amd.h); torch is
only device memory and streams.
"""
import math
from collections import namedtuple
from typing import Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _capi
from .params import FlowConfig, Params, flatten

Flow = namedtuple("Flow", [
  "log_prob", "sample", "sample_and_log_prob", "forward", "inverse",
  "forward_jac", "inverse_jac", "gauge_potential",
])

_MASK64 = (1 << 64) - 1

# In-place writers that bypass torch's version counter (Adam.apply through the C ABI) bump the epoch of the
# memory they wrote; FlowEngine.load compares it.  Keyed by data_ptr: an entry can only go stale when the memory
# is recycled, which costs a spurious re-preparation, never a missed one (the engine keeps its tensor alive).
_WRITE_EPOCH = {}


def mark_updated(flat: torch.Tensor) -> None:
  _WRITE_EPOCH[flat.data_ptr()] = _WRITE_EPOCH.get(flat.data_ptr(), 0) + 1


def _c_config(cfg: FlowConfig) -> _capi.CnfConfig:
  return _capi.CnfConfig(cfg.dim, cfg.num_layers, cfg.hidden_size, cfg.mlp_num_layers,
                         cfg.num_bins, cfg.range_min, cfg.range_max, cfg.min_bin_size,
                         cfg.min_knot_slope, 1 if cfg.periodized else 0)


def _stream_ptr(device) -> int:
  """hipStream_t of torch's current stream on `device` (the raw handle: no Stream object per call)."""
  idx = device.index if isinstance(device, torch.device) else torch.device(device).index
  if idx is None:
    idx = torch.cuda.current_device()
  return torch._C._cuda_getCurrentRawStream(idx)


class _OnDevice:
  """`with torch.cuda.device(dev)` for the C calls (hipcc's runtime acts on the CURRENT device), without the cost of
  the context manager when `dev` already is current -- every launch of a loss evaluation goes through one of these."""
  __slots__ = ("dev", "ctx")

  def __init__(self, dev):
    self.dev, self.ctx = dev, None

  def __enter__(self):
    if torch.cuda.current_device() != self.dev.index:
      self.ctx = torch.cuda.device(self.dev)
      self.ctx.__enter__()
    return self

  def __exit__(self, *exc):
    if self.ctx is not None:
      self.ctx.__exit__(*exc)
      self.ctx = None
    return False


def seed_to_u64(seed) -> Tuple[int, int]:
  """int | (seed, first_sample) | 2-word PRNGKey-like array -> (u64 seed, first_sample)."""
  offset = 0
  if isinstance(seed, tuple) and len(seed) == 2 and not torch.is_tensor(seed[0]):
    seed, offset = seed
  if torch.is_tensor(seed) or isinstance(seed, np.ndarray):
    words = [int(v) for v in np.asarray(seed.cpu() if torch.is_tensor(seed) else seed).reshape(-1)]
    if len(words) == 1:
      seed = words[0]
    elif len(words) == 2:   # jax.random.PRNGKey layout: two uint32 words
      seed = ((words[0] & 0xFFFFFFFF) << 32) | (words[1] & 0xFFFFFFFF)
    else:
      raise ValueError("seed array must have 1 or 2 words")
  return int(seed) & _MASK64, int(offset)


_MULTI_TYPES = {}        # n_terms -> the ctypes array types of cnf_loss_terms_grad_multi's arguments


class DeviceRng:
  """A training step's random key in DEVICE memory: `state` = int64[2] = [step count, key] (include/cnf_ot_amd.h,
  "a training step as one device-side program").  Pass it wherever the loss functions take `rng`: base noise, the time
  batch and the mixture components are then drawn by kernels that read the key on the device, so a step can be
  captured into a HIP graph once and replayed with a new key -- `set_key` is one 8-byte asynchronous copy."""

  SLOTS = 32       # keys in flight: the host may run this many steps ahead of the GPU before it waits for a copy

  def __init__(self, device):
    self.device = torch.device(device)
    self.state = torch.zeros(2, dtype=torch.int64, device=self.device)
    self._pin = torch.zeros(self.SLOTS, dtype=torch.int64).pin_memory()
    self._done = [None] * self.SLOTS          # event behind each slot's last copy
    self._n = 0

  def set_key(self, rng) -> "DeviceRng":
    seed, off = seed_to_u64(rng)
    seed = (seed + 0x9E3779B97F4A7C15 * off) & _MASK64
    k = self._n % self.SLOTS
    self._n += 1
    if self._done[k] is not None:
      self._done[k].synchronize()             # (the asynchronous copy reads the pinned slot when it EXECUTES)
    else:
      self._done[k] = torch.cuda.Event()
    self._pin[k] = seed - (1 << 64) if seed >= (1 << 63) else seed
    self.state[1:2].copy_(self._pin[k:k + 1], non_blocking=True)
    self._done[k].record(torch.cuda.current_stream(self.device))
    return self

  @property
  def ptr(self) -> int:
    return self.state.data_ptr()


class FlowEngine:
  """One CnfModel handle (C ABI) on one GPU.  Explicit, allocation-light API
  used by the loss code, the evaluators and bench.py."""

  def __init__(self, cfg: FlowConfig, device: Union[str, torch.device, int] = "cuda"):
    self.lib = _capi.lib()            # raises if the HIP library is missing
    self.cfg = cfg
    self.device = torch.device(device)
    if self.device.type != "cuda":
      raise ValueError("cnf_ot_amd runs on MI355X only: pass a cuda (ROCm) device")
    if self.device.index is None:
      self.device = torch.device("cuda", torch.cuda.current_device())
    ccfg = _c_config(cfg)
    handle = _capi.ctypes.c_void_p()
    with _OnDevice(self.device):
      _capi.check(self.lib.cnf_model_create(_capi.ctypes.byref(ccfg), _capi.ctypes.byref(handle)),
                  "cnf_model_create")
    self._h = handle
    self._flat = None                 # keeps the caller's flat tensor alive
    self._flat_key = None
    self._reserved = {}               # stream -> table sets reserved (cnf_model_reserve)
    self._pwl_mode = 1
    self._precise = True
    self._tables_ok = (cfg.dim == 2 and cfg.hidden_size == 16 and cfg.num_bins == 5 and cfg.mlp_num_layers == 2
                       and not cfg.periodized)

  def __del__(self):
    h = getattr(self, "_h", None)
    if h:
      try:
        self.lib.cnf_model_destroy(h)
      except Exception:
        pass
      self._h = None

  # -- parameters ------------------------------------------------------------
  def load(self, params, assume_unchanged: bool = False) -> "FlowEngine":
    """cnf_model_set_params: prepare the `first` table + snapshot the weights -- on EVERY call by default (one small
    kernel on the stream), like passing `params` to a pure function.
    assume_unchanged=True (opt-in; `FlowModel(assume_unchanged_params=True)` turns it on for model.apply.*): skip the
    preparation when `params` is the very tensor that is already loaded and nothing visible has written to it since --
    same address, same torch version counter, same write epoch (`mark_updated`, bumped by Adam.apply).  NOT every
    in-place writer is visible this way: torch.distributed collectives (all_reduce, broadcast), writes through
    `.data` and kernels of other libraries leave the version counter alone -- after those call `mark_updated(flat)`
    (cnf_ot_amd.distributed.broadcast_params / all_reduce_params do) or load without the flag."""
    flat = flatten(self.cfg, params, self.device)
    # the same memory (this engine keeps `_flat` alive, so its address cannot have been recycled), same torch
    # version (aliases made by detach() share the counter), no C-side write since
    key = (flat.data_ptr(), flat._version, _WRITE_EPOCH.get(flat.data_ptr(), 0))
    if assume_unchanged and self._flat is not None and key == self._flat_key:
      return self
    with _OnDevice(self.device):
      _capi.check(self.lib.cnf_model_set_params(self._h, flat.data_ptr(), _stream_ptr(self.device)),
                  "cnf_model_set_params")
    self._flat = flat
    self._flat_key = key
    return self

  def reserve(self, n_slices: int, sets_per_slice: int = 1) -> int:
    """cnf_model_reserve for the current stream: room for the conditioner
    tables of `n_slices` time-slices (x sets_per_slice conditions; a call with
    more slices than 2 048 is processed in chunks of that).  Allocates, so it is
    skipped while the stream is being captured into a graph (the call then uses
    what is reserved, or the MLP kernels).  Returns the sets now reserved."""
    if not self._tables_ok:
      return 0
    stream = _stream_ptr(self.device)
    have = self._reserved.get(stream, 0)
    want = min(int(n_slices), 2048) * int(sets_per_slice)
    if want <= have or torch.cuda.is_current_stream_capturing():
      return have
    want = max(16, 1 << (want - 1).bit_length())          # grow geometrically: few re-allocations
    with _OnDevice(self.device):
      _capi.check(self.lib.cnf_model_reserve(self._h, stream, want), "cnf_model_reserve")
    self._reserved[stream] = want
    return want

  def last_path(self) -> str:
    """Which kernels the most recent compute call ran (cnf_model_last_path)."""
    return _capi.PATH_NAMES.get(self.lib.cnf_model_last_path(self._h), "?")

  def set_profiling(self, on: bool) -> None:
    _capi.check(self.lib.cnf_model_set_profiling(self._h, 1 if on else 0), "cnf_model_set_profiling")

  def read_profile(self):
    """(flow kernel ms, table build ms, launches, samples) summed over the
    profiled launches since the last read; waits for them."""
    f, b = _capi.ctypes.c_double(), _capi.ctypes.c_double()
    n, smp = _capi.ctypes.c_int64(), _capi.ctypes.c_int64()
    _capi.check(self.lib.cnf_model_read_profile(self._h, _capi.ctypes.byref(f), _capi.ctypes.byref(b),
                                                _capi.ctypes.byref(n), _capi.ctypes.byref(smp)),
                "cnf_model_read_profile")
    return f.value, b.value, n.value, smp.value

  def set_fast_math(self, on: bool) -> None:
    _capi.check(self.lib.cnf_model_set_fast_math(self._h, 1 if on else 0), "cnf_model_set_fast_math")

  def set_mfma(self, mode) -> None:
    """MFMA (v_mfma_f32_16x16x4_f32) conditioner (hidden 16, 5 bins, fast
    math): True / 1 = wherever available, False / 0 = packed-VALU conditioner,
    2 = MFMA for launches that leave the chip under-filled (the default)."""
    _capi.check(self.lib.cnf_model_set_mfma(self._h, int(mode)), "cnf_model_set_mfma")

  def set_pwl(self, mode: int) -> None:
    """Piecewise-linear conditioner tables (dim 2, slice-uniform condition):
    1 = for large launches (default), 2 = whenever they apply, 0 = never."""
    _capi.check(self.lib.cnf_model_set_pwl(self._h, int(mode)), "cnf_model_set_pwl")
    self._pwl_mode = int(mode)

  def set_dpar(self, mode: int) -> None:
    """Wave-per-dimension kernel for base -> data at dim >= 3: 1 = chosen by
    batch size (default), 2 = always, 0 = never."""
    _capi.check(self.lib.cnf_model_set_dpar(self._h, int(mode)), "cnf_model_set_dpar")

  def set_precise(self, on: bool) -> None:
    """cnf_model_set_precise: float64 position path of log_prob / inverse
    (default on), or plain fp32."""
    _capi.check(self.lib.cnf_model_set_precise(self._h, 1 if on else 0), "cnf_model_set_precise")
    self._precise = bool(on)

  def set_samples_per_lane(self, spl: int) -> None:
    """0: chosen by batch size (default); 1 / 2: force the one-sample or the
    packed two-samples-per-lane kernel."""
    _capi.check(self.lib.cnf_model_set_samples_per_lane(self._h, int(spl)), "cnf_model_set_samples_per_lane")

  # -- helpers ---------------------------------------------------------------
  def _points(self, t, what, keep_f64=False) -> torch.Tensor:
    """float64 points select the float64 kernels (the reference's dtype:
    exact-mode parity, slower); everything else runs in float32."""
    if torch.is_tensor(t) and t.device == self.device and t.dim() == 2 and t.shape[1] == self.cfg.dim and \
        t.is_contiguous() and (t.dtype == torch.float32 or (keep_f64 and t.dtype == torch.float64)):
      return t                       # (the common case: nothing to convert, no torch call made)
    if not torch.is_tensor(t):
      t = torch.as_tensor(np.asarray(t))
    if t.dim() != 2 or t.shape[1] != self.cfg.dim:
      # the reference checks the event shape at trace time (autoregressive.py:80,113)
      raise ValueError(f"{what}: expected shape [B, {self.cfg.dim}], got {tuple(t.shape)}")
    dtype = torch.float64 if (keep_f64 and t.dtype == torch.float64) else torch.float32
    return t.to(device=self.device, dtype=dtype).contiguous()

  def cond(self, cond, B: int, dtype=torch.float32) -> Tuple[torch.Tensor, int]:
    """cond -> (flat device tensor, c_block).  [B,1]/[B]: per sample;
    scalar/[1]: broadcast; [S]/[S,1] with B % S == 0: S equal slices."""
    if torch.is_tensor(cond) and cond.device == self.device and cond.dtype == dtype and cond.is_contiguous():
      c = cond if cond.dim() == 1 else cond.reshape(-1)
    else:
      if not torch.is_tensor(cond):
        cond = torch.as_tensor(np.asarray(cond, dtype=np.float64))
      c = cond.to(device=self.device, dtype=dtype).reshape(-1).contiguous()
    n = c.numel()
    if n == 0:
      raise ValueError("cond is empty")
    if n == 1:
      return c, max(B, 1)
    if n == B:
      return c, 1
    if B % n == 0:
      return c, B // n
    raise ValueError(f"cond with {n} values does not tile a batch of {B}")

  def _check_out(self, t, shape, what, dtype=torch.float32):
    if (t.dtype != dtype or t.device != self.device or not t.is_contiguous()
        or tuple(t.shape) != tuple(shape)):
      raise ValueError(f"{what}: need a contiguous {dtype} {tuple(shape)} tensor on {self.device}")
    return t

  def _run(self, fn, name, pts, cond, want_pts, want_aux, out=None, aux=None):
    B = pts.shape[0]
    if pts.dtype == torch.float64:
      fn, name = getattr(self.lib, name + "_f64"), name + "_f64"
    c, c_block = self.cond(cond, B, pts.dtype)
    if self._pwl_mode and B > 0 and pts.dtype == torch.float32:
      self.reserve(-(-B // c_block) if c_block > 1 else 1)      # per-sample cond: one set, if it proves uniform
    if out is not None:
      self._check_out(out, pts.shape, name + " out", pts.dtype)
    elif want_pts:
      out = torch.empty_like(pts)
    if aux is not None:
      self._check_out(aux, (B,), name + " aux", pts.dtype)
    elif want_aux:
      aux = torch.empty(B, dtype=pts.dtype, device=self.device)
    if B > 0:
      with _OnDevice(self.device):
        _capi.check(fn(self._h, pts.data_ptr(), c.data_ptr(), c_block,
                       out.data_ptr() if out is not None else None,
                       aux.data_ptr() if aux is not None else None, B,
                       _stream_ptr(self.device)), name)
    return out, aux

  # -- the C ABI, on tensors -------------------------------------------------
  def forward_logdet(self, x, cond, want_logdet=True):
    """base -> data: (y [B,D], log|det J| [B])."""
    x = self._points(x, "forward", keep_f64=True)
    return self._run(self.lib.cnf_forward_logdet, "cnf_forward_logdet", x, cond, True, want_logdet)

  def inverse_logdet(self, y, cond, want_logdet=True):
    """data -> base: (x [B,D], log|det J^-1| [B])."""
    y = self._points(y, "inverse", keep_f64=True)
    return self._run(self.lib.cnf_inverse_logdet, "cnf_inverse_logdet", y, cond, True, want_logdet)

  def log_prob(self, value, cond) -> torch.Tensor:
    value = self._points(value, "log_prob", keep_f64=True)
    B = value.shape[0]
    f64 = value.dtype == torch.float64
    c, c_block = self.cond(cond, B, value.dtype)
    if self._pwl_mode and B > 0 and not f64:
      self.reserve(-(-B // c_block) if c_block > 1 else 1)
    lp = torch.empty(B, dtype=value.dtype, device=self.device)
    if B > 0:
      fn = self.lib.cnf_log_prob_f64 if f64 else self.lib.cnf_log_prob
      with _OnDevice(self.device):
        _capi.check(fn(self._h, value.data_ptr(), c.data_ptr(), c_block, lp.data_ptr(), B,
                       _stream_ptr(self.device)), "cnf_log_prob")
    return lp

  def sample_logprob(self, noise, cond, want_logp=True, out=None, logp_out=None):
    """(samples [B,D], log_prob [B]) from base noise [B,D].  `out` / `logp_out`
    are optional preallocated result tensors (no allocation on the call)."""
    noise = self._points(noise, "sample", keep_f64=True)
    return self._run(self.lib.cnf_sample_logprob, "cnf_sample_logprob", noise, cond, True, want_logp,
                     out=out, aux=logp_out)

  def sample_logprob_seeded(self, seed, n_samples: int, cond, want_logp=True, first_sample: int = 0,
                            slice_stride: Optional[int] = None, out=None, logp_out=None):
    """cnf_sample_logprob_seeded: `sample_logprob(self.normal(seed, n_samples, first_sample), cond)` bit for bit, the
    base noise drawn inside the flow kernel (no noise tensor; one launch).  slice_stride: stream samples between
    the slices of `cond` (default: the slice length -- n_samples consecutive samples; 0: every slice the same draw)."""
    seed, off = seed_to_u64(seed)
    B = int(n_samples)
    c, c_block = self.cond(cond, B)
    if slice_stride is None:
      slice_stride = min(c_block, B) if c_block > 1 else 1
    if self._pwl_mode and B > 0:
      self.reserve(-(-B // c_block) if c_block > 1 else 1)
    D = self.cfg.dim
    if out is not None:
      self._check_out(out, (B, D), "sample_logprob_seeded out")
    else:
      out = torch.empty(B, D, dtype=torch.float32, device=self.device)
    if logp_out is not None:
      self._check_out(logp_out, (B,), "sample_logprob_seeded logp")
    elif want_logp:
      logp_out = torch.empty(B, dtype=torch.float32, device=self.device)
    if B > 0:
      with _OnDevice(self.device):
        _capi.check(self.lib.cnf_sample_logprob_seeded(self._h, seed, off + int(first_sample), int(slice_stride),
                                                       c.data_ptr(), c_block, out.data_ptr(),
                                                       logp_out.data_ptr() if logp_out is not None else None, B,
                                                       _stream_ptr(self.device)), "cnf_sample_logprob_seeded")
    return out, logp_out

  def slice_conds(self, t) -> torch.Tensor:
    """[n_slices] float32 on the device.  Host lists are uploaded once and kept (a loss evaluation asks for the
    same few condition lists -- [0], [T], the step's time batch -- for every term, loss and gradient alike)."""
    if torch.is_tensor(t):
      if t.device == self.device and t.dtype == torch.float32 and t.is_contiguous():
        return t if t.dim() == 1 else t.reshape(-1)
      return t.to(device=self.device, dtype=torch.float32).reshape(-1).contiguous()
    a = np.ascontiguousarray(np.asarray(t, dtype=np.float32).reshape(-1))
    key = a.tobytes()
    cache = self.__dict__.setdefault("_cond_cache", {})
    d = cache.get(key)
    if d is None:
      if len(cache) >= 32:
        cache.clear()
      d = cache[key] = torch.from_numpy(a.copy()).to(self.device)
    return d

  def loss_terms(self, spec: "_capi.CnfLossSpec", pts, t, B: int, shared: bool, sums=None) -> torch.Tensor:
    """cnf_loss_terms: per-slice SUMS (float64 [n_slices]) of one Monte-Carlo
    loss term.  pts: base noise (or data points) [B, D] if `shared` else
    [n_slices*B, D]; t: [n_slices]."""
    pts = self._points(pts, "loss_terms")
    t = self.slice_conds(t)
    n_slices = t.numel()
    need = B if shared else n_slices * B
    if pts.shape[0] != need:
      raise ValueError(f"loss_terms: pts has {pts.shape[0]} rows, expected {need}")
    if sums is None:
      sums = torch.empty(n_slices, dtype=torch.float64, device=self.device)
    if self._pwl_mode and n_slices > 0:
      self.reserve(n_slices, _sets_of(spec))
    if n_slices > 0:
      with _OnDevice(self.device):
        _capi.check(self.lib.cnf_loss_terms(self._h, _capi.ctypes.byref(spec), pts.data_ptr(),
                                            1 if shared else 0, t.data_ptr(), n_slices, B,
                                            sums.data_ptr(), _stream_ptr(self.device)), "cnf_loss_terms")
    return sums

  def loss_terms_seeded(self, spec, seed, t, B: int, first_sample: int = 0, slice_stride: int = 0) -> torch.Tensor:
    """cnf_loss_terms_seeded: as `loss_terms`, base noise drawn in the kernel
    (sample i of slice s = stream sample first_sample + s*slice_stride + i)."""
    seed, off = seed_to_u64(seed)
    t = self.slice_conds(t)
    sums = torch.empty(t.numel(), dtype=torch.float64, device=self.device)
    if self._pwl_mode and t.numel() > 0:
      self.reserve(t.numel(), _sets_of(spec))
    if t.numel() > 0:
      with _OnDevice(self.device):
        _capi.check(self.lib.cnf_loss_terms_seeded(self._h, _capi.ctypes.byref(spec), seed, off + first_sample,
                                                   slice_stride, t.data_ptr(), t.numel(), B, sums.data_ptr(),
                                                   _stream_ptr(self.device)), "cnf_loss_terms_seeded")
    return sums

  def loss_terms_grad(self, spec, pts, t, B: int, shared: bool, scale: float, grad: torch.Tensor, sums=None) -> torch.Tensor:
    """cnf_loss_terms_grad: like `loss_terms`, and accumulates
    scale * d(sum of the term)/d(params) into `grad` (flat float32 [n_params])."""
    if self._flat is None:
      raise RuntimeError("load(params) before asking for gradients")
    if not getattr(self, "_grad_enabled", False):
      with _OnDevice(self.device):
        _capi.check(self.lib.cnf_grad_enable(self._h, 0), "cnf_grad_enable")
      self._grad_enabled = True
    pts = self._points(pts, "loss_terms_grad")
    t = self.slice_conds(t)
    n_slices = t.numel()
    need = B if shared else n_slices * B
    if pts.shape[0] != need:
      raise ValueError(f"loss_terms_grad: pts has {pts.shape[0]} rows, expected {need}")
    self._check_out(grad, (self.cfg.param_count(),), "grad")
    if sums is None:
      sums = torch.empty(n_slices, dtype=torch.float64, device=self.device)
    if n_slices > 0:
      with _OnDevice(self.device):
        _capi.check(self.lib.cnf_loss_terms_grad(self._h, _capi.ctypes.byref(spec), pts.data_ptr(),
                                                 1 if shared else 0, t.data_ptr(), n_slices, B, float(scale),
                                                 sums.data_ptr(), grad.data_ptr(), self._flat.data_ptr(),
                                                 _stream_ptr(self.device)), "cnf_loss_terms_grad")
    return sums

  def loss_terms_grad_multi(self, jobs, grad: torch.Tensor) -> None:
    """cnf_loss_terms_grad_multi: several terms of one loss in ONE launch.  jobs: (spec, pts, t, B, shared, scale, sums)
    tuples, at most 4 per launch (longer lists go out in groups); `sums` tensors are filled, `grad` accumulated."""
    if self._flat is None:
      raise RuntimeError("load(params) before asking for gradients")
    if not getattr(self, "_grad_enabled", False):
      with _OnDevice(self.device):
        _capi.check(self.lib.cnf_grad_enable(self._h, 0), "cnf_grad_enable")
      self._grad_enabled = True
    self._check_out(grad, (self.cfg.param_count(),), "grad")
    for g0 in range(0, len(jobs), 4):
      grp = jobs[g0:g0 + 4]
      n = len(grp)
      T = _MULTI_TYPES.get(n)
      if T is None:        # (ctypes array TYPES are built on first use: creating them per call cost more than the call)
        C = _capi.ctypes
        T = _MULTI_TYPES[n] = (_capi.CnfLossSpec * n, C.c_void_p * n, C.c_int32 * n, C.c_int64 * n, C.c_float * n)
      specs, pts_a, t_a, sums_a = T[0](), T[1](), T[1](), T[1]()
      shared_a, ns_a, B_a, sc_a = T[2](), T[3](), T[3](), T[4]()
      keep = []
      for i, (spec, pts, t, B, shared, scale, sums) in enumerate(grp):
        pts = self._points(pts, "loss_terms_grad_multi")
        t = self.slice_conds(t)
        nt = t.numel()
        need = B if shared else nt * B
        if pts.shape[0] != need:
          raise ValueError(f"loss_terms_grad_multi: pts has {pts.shape[0]} rows, expected {need}")
        if sums.numel() != nt or sums.dtype != torch.float64:
          raise ValueError("loss_terms_grad_multi: sums must be float64 [n_slices]")
        keep += [pts, t]
        specs[i] = spec
        pts_a[i], t_a[i], sums_a[i] = pts.data_ptr(), t.data_ptr(), sums.data_ptr()
        shared_a[i], ns_a[i], B_a[i], sc_a[i] = 1 if shared else 0, nt, int(B), float(scale)
      with _OnDevice(self.device):
        _capi.check(self.lib.cnf_loss_terms_grad_multi(self._h, n, specs, pts_a, shared_a, t_a, ns_a, B_a, sc_a, sums_a,
                                                       grad.data_ptr(), self._flat.data_ptr(), _stream_ptr(self.device)),
                    "cnf_loss_terms_grad_multi")

  def input_vjp(self, pts, cond, ybar=None, ldbar=None, to_base=False) -> torch.Tensor:
    """cnf_input_vjp: xbar = ybar . dF/dx + ldbar * d logdet/dx of one flow pass."""
    pts = self._points(pts, "input_vjp")
    B = pts.shape[0]
    c, c_block = self.cond(cond, B)
    if ybar is not None:
      ybar = self._check_out(self._points(ybar, "ybar"), pts.shape, "ybar")
    if ldbar is not None:
      ldbar = ldbar.to(device=self.device, dtype=torch.float32).reshape(-1).contiguous()
      if ldbar.numel() != B:
        raise ValueError("ldbar must have one value per sample")
    xbar = torch.empty_like(pts)
    if B > 0:
      with _OnDevice(self.device):
        _capi.check(self.lib.cnf_input_vjp(self._h, 1 if to_base else 0, pts.data_ptr(), c.data_ptr(), c_block,
                                           ybar.data_ptr() if ybar is not None else None,
                                           ldbar.data_ptr() if ldbar is not None else None,
                                           xbar.data_ptr(), B, _stream_ptr(self.device)), "cnf_input_vjp")
    return xbar

  def pass_vjp(self, pts, cond, ybar, ldbar, to_base, grad=None, want_xbar=True):
    """cnf_pass_vjp: input adjoints (returned, or None) and, into `grad`, the
    parameter gradient of one flow pass for the output adjoints (ybar, ldbar)."""
    pts = self._points(pts, "pass_vjp")
    B = pts.shape[0]
    c, c_block = self.cond(cond, B)
    ybar = None if ybar is None else self._check_out(self._points(ybar, "ybar"), pts.shape, "ybar")
    ldbar = None if ldbar is None else ldbar.to(device=self.device, dtype=torch.float32).reshape(-1).contiguous()
    if grad is not None:
      if self._flat is None:
        raise RuntimeError("load(params) before asking for gradients")
      if not getattr(self, "_grad_enabled", False):
        with _OnDevice(self.device):
          _capi.check(self.lib.cnf_grad_enable(self._h, 0), "cnf_grad_enable")
        self._grad_enabled = True
      self._check_out(grad, (self.cfg.param_count(),), "grad")
      if self._pwl_mode and B > 0:          # the table form of the backward builds tables for chunks of 128 slices
        self.reserve(min(-(-B // c_block), 128))
    xbar = torch.empty_like(pts) if want_xbar else None
    if B > 0:
      with _OnDevice(self.device):
        _capi.check(self.lib.cnf_pass_vjp(self._h, 1 if to_base else 0, pts.data_ptr(), c.data_ptr(), c_block,
                                          ybar.data_ptr() if ybar is not None else None,
                                          ldbar.data_ptr() if ldbar is not None else None,
                                          xbar.data_ptr() if xbar is not None else None,
                                          grad.data_ptr() if grad is not None else None,
                                          self._flat.data_ptr() if grad is not None else None, B,
                                          _stream_ptr(self.device)), "cnf_pass_vjp")
    return xbar

  def neg_logprob_vjp(self, pts, cond, loss_coef: float, grad, sums=None):
    """cnf_neg_logprob_vjp: per-slice sums of -log_prob(pts; cond) (float64 device tensor) and, into `grad`,
    loss_coef * d(their total) / d(params) -- one launch over the data.  Returns None where the table backward does not
    apply (CNF_ERR_UNSUPPORTED): the caller composes the term from inverse_logdet + term_residual + pass_vjp."""
    pts = self._points(pts, "neg_logprob_vjp")
    B = pts.shape[0]
    c, c_block = self.cond(cond, B)
    if self._flat is None:
      raise RuntimeError("load(params) before asking for gradients")
    if not getattr(self, "_grad_enabled", False):
      with _OnDevice(self.device):
        _capi.check(self.lib.cnf_grad_enable(self._h, 0), "cnf_grad_enable")
      self._grad_enabled = True
    self._check_out(grad, (self.cfg.param_count(),), "grad")
    n_slices = max(-(-B // c_block), 1)
    if sums is None:
      sums = torch.empty(n_slices, dtype=torch.float64, device=self.device)
    if not self._pwl_mode or B == 0:
      return None
    self.reserve(min(n_slices, 128))
    with _OnDevice(self.device):
      rc = self.lib.cnf_neg_logprob_vjp(self._h, pts.data_ptr(), c.data_ptr(), c_block, float(loss_coef), sums.data_ptr(),
                                        grad.data_ptr(), self._flat.data_ptr(), B, _stream_ptr(self.device))
    if rc == _capi.CNF_ERR_UNSUPPORTED:
      return None
    _capi.check(rc, "cnf_neg_logprob_vjp")
    return sums

  def kinetic_potential_vjp(self, z, conds, S: int, dt: float, c_kin: float, grad, subtype: int = -1, a: float = 0.0,
                            c_pot: float = 0.0, kin=None, pot=None):
    """cnf_kinetic_potential_vjp: per-time sums of the kinetic (and, subtype >= 0, the potential) term for the ONE
    draw z [count, 2] pushed to the 2 S (3 S) conditions `conds`, and their gradient into `grad` (None: the values
    alone).  Returns (kin, pot) or None where the call does not apply (the caller composes the term from its parts)."""
    z = self._points(z, "kinetic_potential_vjp")
    count = z.shape[0]
    sets = 3 if subtype >= 0 else 2
    c = conds if (torch.is_tensor(conds) and conds.device == self.device and conds.dtype == torch.float32
                  and conds.is_contiguous()) else torch.as_tensor(conds, dtype=torch.float32, device=self.device).contiguous()
    if c.numel() != sets * S:
      raise ValueError(f"kinetic_potential_vjp: {sets * S} conditions expected, got {c.numel()}")
    if not self._pwl_mode or count == 0 or self.cfg.dim != 2:
      return None
    if grad is not None:
      if self._flat is None:
        raise RuntimeError("load(params) before asking for gradients")
      if not getattr(self, "_grad_enabled", False):
        with _OnDevice(self.device):
          _capi.check(self.lib.cnf_grad_enable(self._h, 0), "cnf_grad_enable")
        self._grad_enabled = True
      self._check_out(grad, (self.cfg.param_count(),), "grad")
    if sets * S > 128:
      return None
    self.reserve(sets * S)
    kin = torch.empty(S, dtype=torch.float64, device=self.device) if kin is None else kin
    if subtype >= 0 and pot is None:
      pot = torch.empty(S, dtype=torch.float64, device=self.device)
    need = 4 * sets * S * count
    work = getattr(self, "_kp_work", None)
    if work is None or work.numel() < need:
      work = self._kp_work = torch.empty(need, dtype=torch.float32, device=self.device)
    with _OnDevice(self.device):
      rc = self.lib.cnf_kinetic_potential_vjp(self._h, z.data_ptr(), count, c.data_ptr(), int(S), float(dt), float(c_kin),
                                              int(subtype), float(a), float(c_pot), kin.data_ptr(),
                                              pot.data_ptr() if subtype >= 0 else None,
                                              grad.data_ptr() if grad is not None else None,
                                              self._flat.data_ptr() if grad is not None else None, work.data_ptr(),
                                              _stream_ptr(self.device))
    if rc == _capi.CNF_ERR_UNSUPPORTED:
      return None
    _capi.check(rc, "cnf_kinetic_potential_vjp")
    return kin, pot

  def logprob_fd(self, pts, cond, dx: float) -> torch.Tensor:
    """cnf_logprob_fd: the central-difference score [B, D] of log_prob at pts."""
    pts = self._points(pts, "logprob_fd")
    B = pts.shape[0]
    c, c_block = self.cond(cond, B)
    score = torch.empty_like(pts)
    if B > 0:
      with _OnDevice(self.device):
        _capi.check(self.lib.cnf_logprob_fd(self._h, pts.data_ptr(), c.data_ptr(), c_block, float(dx),
                                            score.data_ptr(), B, _stream_ptr(self.device)), "cnf_logprob_fd")
    return score

  def logprob_fd_vjp(self, pts, cond, dx: float, gbar, grad, want_pts_bar=True):
    """cnf_logprob_fd_vjp: backward of `logprob_fd` for the output adjoint gbar
    [B, D]: returns pts_bar (or None) and accumulates the parameter gradient
    into `grad`."""
    if self._flat is None:
      raise RuntimeError("load(params) before asking for gradients")
    if not getattr(self, "_grad_enabled", False):
      with _OnDevice(self.device):
        _capi.check(self.lib.cnf_grad_enable(self._h, 0), "cnf_grad_enable")
      self._grad_enabled = True
    pts = self._points(pts, "logprob_fd_vjp")
    B = pts.shape[0]
    c, c_block = self.cond(cond, B)
    gbar = self._check_out(self._points(gbar, "gbar"), pts.shape, "gbar")
    self._check_out(grad, (self.cfg.param_count(),), "grad")
    pts_bar = torch.empty_like(pts) if want_pts_bar else None
    if B > 0:
      with _OnDevice(self.device):
        _capi.check(self.lib.cnf_logprob_fd_vjp(self._h, pts.data_ptr(), c.data_ptr(), c_block, float(dx),
                                                gbar.data_ptr(), pts_bar.data_ptr() if want_pts_bar else None,
                                                grad.data_ptr(), self._flat.data_ptr(), B,
                                                _stream_ptr(self.device)), "cnf_logprob_fd_vjp")
    return pts_bar

  def score_fd_vjp(self, r, cond, count: int, dt: float, dx: float, coef: float, drift: int, a: float, loss_coef: float,
                   grad: torch.Tensor):
    """cnf_score_fd_vjp: per-slice sums (float64 [n / count]) of the score-term residual from r [3n, D] AND its
    backward in one launch: returns (sums, rbar [3n, D]); the parameter gradient is accumulated into `grad`."""
    if self._flat is None:
      raise RuntimeError("load(params) before asking for gradients")
    if not getattr(self, "_grad_enabled", False):
      with _OnDevice(self.device):
        _capi.check(self.lib.cnf_grad_enable(self._h, 0), "cnf_grad_enable")
      self._grad_enabled = True
    r = self._points(r, "score_fd_vjp")
    n = r.shape[0] // 3
    if r.shape[0] != 3 * n or n % count:
      raise ValueError("r must hold the samples at t - dt/2 | t + dt/2 | t of whole slices")
    c = self.slice_conds(cond)
    if c.numel() != n // count:
      raise ValueError("one condition per slice")
    self._check_out(grad, (self.cfg.param_count(),), "grad")
    sums = torch.empty(n // count, dtype=torch.float64, device=self.device)
    rbar = torch.empty_like(r)
    if n > 0:
      with _OnDevice(self.device):
        _capi.check(self.lib.cnf_score_fd_vjp(self._h, r.data_ptr(), c.data_ptr(), int(count), float(dt), float(dx),
                                              float(coef), int(drift), float(a), float(loss_coef), sums.data_ptr(),
                                              rbar.data_ptr(), grad.data_ptr(), self._flat.data_ptr(), n,
                                              _stream_ptr(self.device)), "cnf_score_fd_vjp")
    return sums, rbar

  def score_residual(self, r, score, count: int, dt: float, coef: float, drift: int, a: float,
                     loss_coef: float = 0.0, want_adjoints: bool = False):
    """cnf_score_residual: per-slice sums (float64 [n / count]) of the score-term residual from r [3n, D]
    (samples at t -+ dt/2 and t) and score [n, D]; with want_adjoints also (rbar [3n, D], sbar [n, D])."""
    n, D = score.shape
    sums = torch.empty(-(-n // count), dtype=torch.float64, device=self.device)
    rbar = torch.empty_like(r) if want_adjoints else None
    sbar = torch.empty_like(score) if want_adjoints else None
    with _OnDevice(self.device):
      _capi.check(self.lib.cnf_score_residual(r.data_ptr(), score.data_ptr(), n, count, D, float(dt), float(coef),
                                              int(drift), float(a), float(loss_coef), sums.data_ptr(),
                                              rbar.data_ptr() if want_adjoints else None,
                                              sbar.data_ptr() if want_adjoints else None,
                                              _stream_ptr(self.device)), "cnf_score_residual")
    return sums, rbar, sbar

  def term_residual(self, kind: int, r, aux, count: int, subtype: int = 0, p0: float = 0.0, loss_coef: float = 0.0,
                    want_adjoints: bool = True, rbar_out=None):
    """cnf_term_residual: per-slice sums (float64) of a kinetic / potential / density-fit term from the outputs of
    its flow launches, and the adjoints (rbar, auxbar) of those outputs.  rbar_out: where the adjoints of r go (a
    view of a larger buffer when several terms share one backward launch)."""
    n = r.shape[0] // (2 if kind == _capi.TERM_KINETIC else 1)
    D = r.shape[1]
    sums = torch.empty(-(-n // count), dtype=torch.float64, device=self.device)
    if rbar_out is not None:
      rbar = self._check_out(rbar_out, r.shape, "term_residual rbar")
    else:
      rbar = torch.empty_like(r) if want_adjoints else None
    auxbar = torch.empty_like(aux) if (want_adjoints and aux is not None) else None
    with _OnDevice(self.device):
      _capi.check(self.lib.cnf_term_residual(int(kind), r.data_ptr(), aux.data_ptr() if aux is not None else None, n,
                                             int(count), D, int(subtype), float(p0), float(loss_coef), sums.data_ptr(),
                                             rbar.data_ptr() if rbar is not None else None,
                                             auxbar.data_ptr() if auxbar is not None else None,
                                             _stream_ptr(self.device)), "cnf_term_residual")
    return sums, rbar, auxbar

  def rkl_residual(self, y, lp, t: float, T: float, beta: float, loss_coef: float = 0.0, want_adjoints: bool = False):
    """cnf_rkl_residual: sum_i lp_i - log mixture(y_i) (float64 [1]); with want_adjoints also (ybar, lpbar)."""
    n, D = y.shape
    total = torch.empty(1, dtype=torch.float64, device=self.device)
    ybar = torch.empty_like(y) if want_adjoints else None
    lpbar = torch.empty_like(lp) if want_adjoints else None
    with _OnDevice(self.device):
      _capi.check(self.lib.cnf_rkl_residual(y.data_ptr(), lp.data_ptr(), n, D, float(t), float(T), float(beta),
                                            float(loss_coef), total.data_ptr(),
                                            ybar.data_ptr() if want_adjoints else None,
                                            lpbar.data_ptr() if want_adjoints else None,
                                            _stream_ptr(self.device)), "cnf_rkl_residual")
    return total, ybar, lpbar

  def jacobian(self, pts, cond, to_base=False) -> torch.Tensor:
    """[B, D, D] Jacobian d out_i / d in_j of a flow pass: D vector-Jacobian products."""
    pts = self._points(pts, "jacobian")
    B, D = pts.shape
    J = torch.empty(B, D, D, dtype=torch.float32, device=self.device)
    for i in range(D):
      e = torch.zeros(B, D, dtype=torch.float32, device=self.device)
      e[:, i] = 1.0
      J[:, i, :] = self.input_vjp(pts, cond, ybar=e, to_base=to_base)
    return J

  def normal_threefry(self, key, n_samples: int, first_sample: int = 0, total_samples: int = None,
                      dtype=torch.float32) -> torch.Tensor:
    """cnf_fill_normal_threefry: rows [first_sample, first_sample + n_samples) of
    jax.random.normal(key, (total_samples, D), float64) (classic threefry path), as `dtype`."""
    k0, k1 = jax_key_words(key)
    D = self.cfg.dim
    total = n_samples + first_sample if total_samples is None else total_samples
    out = torch.empty(n_samples, D, dtype=dtype, device=self.device)
    if n_samples > 0:
      f32 = out.data_ptr() if dtype == torch.float32 else None
      f64 = out.data_ptr() if dtype == torch.float64 else None
      if f32 is None and f64 is None:
        raise ValueError("dtype must be float32 or float64")
      with _OnDevice(self.device):
        _capi.check(self.lib.cnf_fill_normal_threefry(k0, k1, total * D, first_sample * D, n_samples * D, f32, f64,
                                                      _stream_ptr(self.device)), "cnf_fill_normal_threefry")
    return out

  def normal(self, seed, n_samples: int, first_sample: int = 0) -> torch.Tensor:
    """Base noise [n_samples, D]: Philox stream element (first_sample+i)*D+d.  `seed`: an integer / key array, or a
    DeviceRng (the key is read on the device: cnf_fill_normal_dev)."""
    D = self.cfg.dim
    if isinstance(seed, DeviceRng):
      out = torch.empty(n_samples, D, dtype=torch.float32, device=self.device)
      if n_samples > 0:
        with _OnDevice(self.device):
          _capi.check(self.lib.cnf_fill_normal_dev(seed.ptr, first_sample * D, n_samples * D, out.data_ptr(),
                                                   _stream_ptr(self.device)), "cnf_fill_normal_dev")
      return out
    seed, off = seed_to_u64(seed)
    out = torch.empty(n_samples, D, dtype=torch.float32, device=self.device)
    if n_samples > 0:
      with _OnDevice(self.device):
        _capi.check(self.lib.cnf_fill_normal(seed, (off + first_sample) * D, n_samples * D,
                                             out.data_ptr(), _stream_ptr(self.device)), "cnf_fill_normal")
    return out


def _sets_of(spec) -> int:
  """table sets per slice of a fused loss term: conditions t -+ dt/2 (and t)."""
  return 2 if spec.kind == _capi.TERM_KINETIC else (3 if spec.kind <= _capi.TERM_FLOW_MATCHING else 1)


def jax_key_words(seed) -> Tuple[int, int]:
  """The two uint32 words of a JAX PRNG key: a 2-word array is taken as is; an int like jax.random.PRNGKey(int)
  ((seed >> 32) & 0xffffffff, seed & 0xffffffff)."""
  if torch.is_tensor(seed) or isinstance(seed, np.ndarray):
    words = [int(v) for v in np.asarray(seed.cpu() if torch.is_tensor(seed) else seed).reshape(-1)]
    if len(words) == 2:
      return words[0] & 0xFFFFFFFF, words[1] & 0xFFFFFFFF
    if len(words) != 1:
      raise ValueError("a JAX key has two uint32 words")
    seed = words[0]
  seed = int(seed)
  return (seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF


def _num_samples(sample_shape) -> Tuple[int, Tuple[int, ...]]:
  if isinstance(sample_shape, (int, np.integer)):
    sample_shape = (int(sample_shape),)
  sample_shape = tuple(int(s) for s in sample_shape)
  return int(np.prod(sample_shape)) if sample_shape else 1, sample_shape


class _Apply:
  """model.apply: pure functions of (params, ...), like hk's Transformed.apply."""

  def __init__(self, model: "FlowModel"):
    self._m = model

  def _engine(self, params, like=None) -> FlowEngine:
    device = None
    if isinstance(params, Params):
      device = params.flat.device
    if (device is None or device.type != "cuda") and torch.is_tensor(like) and like.is_cuda:
      device = like.device
    if device is None or device.type != "cuda":
      device = torch.device("cuda", torch.cuda.current_device())
    return self._m.engine(device).load(params, assume_unchanged=self._m.assume_unchanged_params)

  # conditional.py:316-321
  def log_prob(self, params, value, cond=None):
    if cond is None:
      raise ValueError("log_prob needs `cond` (the flow is conditional, cond_shape=(1,))")
    return self._engine(params, value).log_prob(value, cond)

  def _draw(self, eng, cond, seed, sample_shape, noise):
    n, shape = _num_samples(sample_shape)
    if noise is None:
      if seed is None:
        raise ValueError("sample needs `seed` (or explicit base `noise`)")
      # rng = "threefry": `seed` is a JAX key and the draw is jax.random.normal's (conditional.py:378,399)
      noise = eng.normal_threefry(seed, n) if self._m.rng == "threefry" else eng.normal(seed, n)
    elif tuple(noise.shape) != (n, eng.cfg.dim):
      raise ValueError(f"noise must have shape {(n, eng.cfg.dim)}, got {tuple(noise.shape)}")
    return noise, shape

  def _seeded(self, noise, seed) -> bool:
    """The reference's form -- `seed=` and no explicit noise (conditional.py:376-402) -- with the build's own stream:
    the draw happens inside the flow kernel (cnf_sample_logprob_seeded), same values as drawing first."""
    return noise is None and seed is not None and self._m.rng == "philox"

  # conditional.py:323-351
  def sample(self, params, *, cond, seed=None, sample_shape=(), noise=None):
    eng = self._engine(params, cond if torch.is_tensor(cond) else None)
    if self._seeded(noise, seed):
      n, shape = _num_samples(sample_shape)
      y, _ = eng.sample_logprob_seeded(seed, n, cond, want_logp=False)
      return y.reshape(shape + (eng.cfg.dim,))
    noise, shape = self._draw(eng, cond, seed, sample_shape, noise)
    y, _ = eng.sample_logprob(noise, cond, want_logp=False)
    return y.reshape(shape + (eng.cfg.dim,))

  # conditional.py:353-374
  def sample_and_log_prob(self, params, *, cond, seed=None, sample_shape=(), noise=None):
    eng = self._engine(params, cond if torch.is_tensor(cond) else None)
    if self._seeded(noise, seed):
      n, shape = _num_samples(sample_shape)
      y, lp = eng.sample_logprob_seeded(seed, n, cond, want_logp=True)
      return y.reshape(shape + (eng.cfg.dim,)), lp.reshape(shape)
    noise, shape = self._draw(eng, cond, seed, sample_shape, noise)
    y, lp = eng.sample_logprob(noise, cond, want_logp=True)
    return y.reshape(shape + (eng.cfg.dim,)), lp.reshape(shape)

  # flows.py:221-223: flow.bijector.forward / inverse
  def forward(self, params, x, c):
    return self._engine(params, x).forward_logdet(x, c, want_logdet=False)[0]

  def inverse(self, params, y, c):
    return self._engine(params, y).inverse_logdet(y, c, want_logdet=False)[0]

  # flows.py:203-211: the jax.jacfwd helpers, as vector-Jacobian products of the backward kernels
  def forward_jac(self, params, x, c):
    """vmap(jacfwd(flow.bijector.forward))(x, c): [B, D, D], rows = outputs."""
    return self._engine(params, x).jacobian(x, c, to_base=False)

  def inverse_jac(self, params, y, c):
    """vmap(jacfwd(flow.bijector.inverse))(y, c): [B, D, D]."""
    return self._engine(params, y).jacobian(y, c, to_base=True)

  def gauge_potential(self, params, x, c):
    """jacfwd(x -> log|det J(forward)(x, c)|): the gradient of the log-det,
    [D] for one point like the reference, [B, D] for a batch."""
    single = torch.is_tensor(x) and x.dim() == 1
    xx = x.reshape(1, -1) if single else x
    eng = self._engine(params, xx)
    g = eng.input_vjp(xx, c, ldbar=torch.ones(xx.shape[0], device=eng.device), to_base=False)
    return g[0] if single else g


class FlowModel:
  """What the reference's driver holds after
  ``hk.without_apply_rng(hk.multi_transform(RQSFlow(...)))`` (solvers.py:41-48)."""

  def __init__(self, cfg: FlowConfig, rng: str = "philox", assume_unchanged_params: bool = False):
    if rng not in ("philox", "threefry"):
      raise ValueError("rng must be 'philox' (the build's own stream) or 'threefry' (jax.random.normal's)")
    self.cfg = cfg
    self.rng = rng
    # model.apply.*(params, ...) re-prepares `params` on every call unless this is set (FlowEngine.load)
    self.assume_unchanged_params = bool(assume_unchanged_params)
    self._engines = {}
    a = _Apply(self)
    self.apply = Flow(a.log_prob, a.sample, a.sample_and_log_prob, a.forward, a.inverse,
                      a.forward_jac, a.inverse_jac, a.gauge_potential)

  def engine(self, device) -> FlowEngine:
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
      device = torch.device("cuda", torch.cuda.current_device())
    eng = self._engines.get(device)
    if eng is None:
      eng = self._engines[device] = FlowEngine(self.cfg, device)
    return eng

  def terms_backend(self, params, device=None) -> FlowEngine:
    """The engine that evaluates fused loss terms for `params` (used by
    cnf_ot_amd.applications / cnf_ot_amd.utils)."""
    if device is None:
      device = params.flat.device if isinstance(params, Params) and params.flat.is_cuda else \
        torch.device("cuda", torch.cuda.current_device())
    return self.engine(device).load(params)

  def init(self, rng=0, x=None, c=None, device=None) -> Params:
    """model.init(rng, zeros((1,dim)), zeros((1,))) (solvers.py:54): identity
    flow -- `first` and every linear_out zero (flows.py:48,71-76), hidden
    layers at haiku's default init."""
    if x is not None and (x.shape[-1] != self.cfg.dim):
      raise ValueError(f"init: event dimension {x.shape[-1]} != {self.cfg.dim}")
    seed, _ = seed_to_u64(rng)
    if device is None:
      device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cpu"
    return Params.init(self.cfg, seed=seed % (2 ** 63), device=device)


def RQSFlow(
  event_shape: Sequence[int],
  num_layers: int,
  hidden_sizes: Sequence[int],
  num_bins: int,
  periodized: bool = False,
  cond_shape=(1,),
  base_range=(0, 2 * math.pi),
  rng: str = "philox",
) -> FlowModel:
  """Same signature as cnf_ot/models/flows.py:178-186 (+ `rng`: "philox" = the
  build's own counter-based stream, "threefry" = `seed` is a JAX key and the
  base draw of sample / sample_and_log_prob is jax.random.normal's)."""
  if len(tuple(event_shape)) != 1:
    raise ValueError("event_shape must be (dim,)")
  if tuple(cond_shape) != (1,):
    raise NotImplementedError("only cond_shape=(1,) (time-conditioned flow) is supported")
  hidden_sizes = list(hidden_sizes)
  if not hidden_sizes or any(h != hidden_sizes[0] for h in hidden_sizes):
    raise NotImplementedError("hidden_sizes must be [hidden_size] * mlp_num_layers (solvers.py:44)")
  # periodized (flows.py:58-64,127-131): sin / cos features, [0, 2 pi], circular slopes -- the flow functions
  # (log_prob, sample, sample_and_log_prob, forward, inverse); losses and gradients are not built for it
  make = FlowConfig.torus if periodized else FlowConfig
  cfg = make(dim=int(event_shape[0]), num_layers=int(num_layers),
             hidden_size=int(hidden_sizes[0]), mlp_num_layers=len(hidden_sizes),
             num_bins=int(num_bins))
  return FlowModel(cfg, rng=rng)
