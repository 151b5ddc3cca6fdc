"""ctypes binding of oracle/build/libcnf_oracle.so (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "libcnf_oracle.so")
_lib = None


class _Cfg(ctypes.Structure):
  _fields_ = [
    ("D", ctypes.c_int32), ("L", ctypes.c_int32), ("H", ctypes.c_int32),
    ("M", ctypes.c_int32), ("K", ctypes.c_int32),
    ("range_min", ctypes.c_double), ("range_max", ctypes.c_double),
    ("min_bin_size", ctypes.c_double), ("min_knot_slope", ctypes.c_double),
    ("periodized", ctypes.c_int32),
  ]


@dataclass(frozen=True)
class OracleConfig:
  """Mirror of the reference's network config (config/mfc.yaml:29-33 and the
  constants of cnf_ot/models/flows.py:124-134)."""
  D: int = 2
  L: int = 2
  H: int = 16
  M: int = 2
  K: int = 5
  range_min: float = -10.0
  range_max: float = 10.0
  min_bin_size: float = 1e-4
  min_knot_slope: float = 1e-4
  # flows.py:58-64,127-131: sin/cos features of the conditioner input, boundary_slopes='circular'; the caller sets
  # the range to [0, 2 pi] (OracleConfig.torus)
  periodized: bool = False

  @staticmethod
  def torus(**kw):
    import math
    return OracleConfig(range_min=0.0, range_max=2.0 * math.pi, periodized=True, **kw)

  def c(self):
    return _Cfg(self.D, self.L, self.H, self.M, self.K, self.range_min,
                self.range_max, self.min_bin_size, self.min_knot_slope, 1 if self.periodized else 0)


def build_library(force: bool = False) -> str:
  """Compile the C restatement with the committed Makefile (gcc + OpenMP)."""
  srcs = [os.path.join(_HERE, f)
          for f in ("cnf_oracle.c", "cnf_oracle_impl.h", "cnf_oracle.h")]
  stale = (not os.path.exists(_LIB_PATH)) or any(
    os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
  if force or stale:
    subprocess.run(["make", "-C", _HERE, "-B"] if force else ["make", "-C", _HERE],
                   check=True, capture_output=True)
  return _LIB_PATH


def load_library():
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(_LIB_PATH):
    build_library()
  lib = ctypes.CDLL(_LIB_PATH)
  lib.cnf_oracle_param_count.restype = ctypes.c_size_t
  lib.cnf_oracle_param_count.argtypes = [ctypes.POINTER(_Cfg)]
  lib.cnf_oracle_num_threads.restype = ctypes.c_int
  _lib = lib
  return lib


def set_num_threads(n: int) -> None:
  load_library().cnf_oracle_set_num_threads(ctypes.c_int(int(n)))


def num_threads() -> int:
  return int(load_library().cnf_oracle_num_threads())


def param_count(cfg: OracleConfig) -> int:
  c = cfg.c()
  return int(load_library().cnf_oracle_param_count(ctypes.byref(c)))


def _dt(dtype):
  dtype = np.dtype(dtype)
  if dtype == np.float64:
    return np.float64, "_f64", ctypes.c_double
  if dtype == np.float32:
    return np.float32, "_f32", ctypes.c_float
  raise TypeError(f"oracle supports float64/float32, got {dtype}")


def _ptr(a):
  return a.ctypes.data_as(ctypes.c_void_p)


def _prep(cfg, params, pts, c, dtype):
  npdt, sfx, _ = _dt(dtype)
  params = np.ascontiguousarray(params, dtype=npdt).reshape(-1)
  if params.size != param_count(cfg):
    raise ValueError(f"params has {params.size} values, config needs {param_count(cfg)}")
  pts = np.ascontiguousarray(pts, dtype=npdt)
  if pts.ndim != 2 or pts.shape[1] != cfg.D:
    raise ValueError(f"expected [B,{cfg.D}] points, got {pts.shape}")
  B = pts.shape[0]
  c = np.ascontiguousarray(c, dtype=npdt).reshape(-1)
  if c.size == 1:
    c_block = max(B, 1)
  elif B % c.size == 0:
    c_block = B // c.size
  else:
    raise ValueError(f"cond of {c.size} values does not tile a batch of {B}")
  return npdt, sfx, params, pts, c, c_block, B


def _call2(name, cfg, params, pts, c, dtype, want_second=True):
  npdt, sfx, params, pts, c, c_block, B = _prep(cfg, params, pts, c, dtype)
  out = np.empty_like(pts)
  second = np.empty(B, dtype=npdt)
  cc = cfg.c()
  fn = getattr(load_library(), name + sfx)
  rc = fn(ctypes.byref(cc), _ptr(params), _ptr(pts), _ptr(c),
          ctypes.c_int64(c_block), _ptr(out), _ptr(second), ctypes.c_int64(B))
  if rc != 0:
    raise RuntimeError(f"{name}{sfx} failed with {rc}")
  return out, second


def forward_logdet(cfg, params, x, c, dtype=np.float64):
  """base -> data, (y, log|det J|): flow.bijector.forward_and_log_det."""
  return _call2("cnf_oracle_forward_logdet", cfg, params, x, c, dtype)


def inverse_logdet(cfg, params, y, c, dtype=np.float64):
  """data -> base, (x, log|det J^-1|): flow.bijector.inverse_and_log_det."""
  return _call2("cnf_oracle_inverse_logdet", cfg, params, y, c, dtype)


def sample_logprob(cfg, params, noise, c, dtype=np.float64):
  """(samples, log_prob) from supplied base noise (conditional.py:382-402)."""
  return _call2("cnf_oracle_sample_logprob", cfg, params, noise, c, dtype)


def log_prob(cfg, params, value, c, dtype=np.float64):
  npdt, sfx, params, value, c, c_block, B = _prep(cfg, params, value, c, dtype)
  lp = np.empty(B, dtype=npdt)
  cc = cfg.c()
  rc = getattr(load_library(), "cnf_oracle_log_prob" + sfx)(
    ctypes.byref(cc), _ptr(params), _ptr(value), _ptr(c),
    ctypes.c_int64(c_block), _ptr(lp), ctypes.c_int64(B))
  if rc != 0:
    raise RuntimeError(f"cnf_oracle_log_prob{sfx} failed with {rc}")
  return lp


def rqs(theta, v, K, lo, hi, min_slope, inverse=False, min_bin=1e-4,
        dtype=np.float64):
  """Scalar spline, theta [n,3K+1], v [n] -> (out [n], logdet [n])."""
  npdt, sfx, ct = _dt(dtype)
  theta = np.ascontiguousarray(theta, dtype=npdt).reshape(-1, 3 * K + 1)
  v = np.ascontiguousarray(v, dtype=npdt).reshape(-1)
  assert theta.shape[0] == v.shape[0]
  out = np.empty_like(v)
  ld = np.empty_like(v)
  rc = getattr(load_library(), "cnf_oracle_rqs" + sfx)(
    _ptr(theta), _ptr(v), ctypes.c_int64(v.size), ctypes.c_int(K), ct(lo),
    ct(hi), ct(min_bin), ct(min_slope), ctypes.c_int(1 if inverse else 0),
    _ptr(out), _ptr(ld))
  if rc != 0:
    raise RuntimeError("cnf_oracle_rqs failed")
  return out, ld


def knots(theta, K, lo=-10.0, hi=10.0, min_slope=1e-4, min_bin=1e-4,
          dtype=np.float64):
  npdt, sfx, ct = _dt(dtype)
  theta = np.ascontiguousarray(theta, dtype=npdt).reshape(3 * K + 1)
  xk, yk, dl = (np.empty(K + 1, dtype=npdt) for _ in range(3))
  rc = getattr(load_library(), "cnf_oracle_knots" + sfx)(
    _ptr(theta), ctypes.c_int(K), ct(lo), ct(hi), ct(min_bin), ct(min_slope),
    _ptr(xk), _ptr(yk), _ptr(dl))
  if rc != 0:
    raise RuntimeError("cnf_oracle_knots failed")
  return xk, yk, dl


def normal(seed: int, first_element: int, n: int) -> np.ndarray:
  """Philox4x32-10 + Box-Muller standard normals, float64 evaluation."""
  out = np.empty(n, dtype=np.float64)
  rc = load_library().cnf_oracle_normal_f64(
    ctypes.c_uint64(seed & (2**64 - 1)), ctypes.c_uint64(first_element),
    ctypes.c_int64(n), _ptr(out))
  if rc != 0:
    raise RuntimeError("cnf_oracle_normal_f64 failed")
  return out


def threefry2x32(key, ctr):
  """One Threefry-2x32-20 block: (key0, key1), (ctr0, ctr1) -> (out0, out1)."""
  k = (ctypes.c_uint32 * 2)(*[int(v) & 0xFFFFFFFF for v in key])
  c = (ctypes.c_uint32 * 2)(*[int(v) & 0xFFFFFFFF for v in ctr])
  o = (ctypes.c_uint32 * 2)()
  load_library().cnf_oracle_threefry2x32(k, c, o)
  return int(o[0]), int(o[1])


def normal_threefry(key, size: int, first_element: int = 0, n: int = None) -> np.ndarray:
  """jax.random.normal(key, shape with prod = size, float64), classic threefry path: elements
  [first_element, first_element + n) of the flattened draw."""
  n = size - first_element if n is None else n
  out = np.empty(n, dtype=np.float64)
  rc = load_library().cnf_oracle_normal_threefry_f64(
    ctypes.c_uint32(int(key[0]) & 0xFFFFFFFF), ctypes.c_uint32(int(key[1]) & 0xFFFFFFFF), ctypes.c_uint64(size),
    ctypes.c_uint64(first_element), ctypes.c_int64(n), _ptr(out))
  if rc != 0:
    raise RuntimeError("cnf_oracle_normal_threefry_f64 failed")
  return out
