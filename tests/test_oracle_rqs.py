"""Pins the oracle's scalar spline (CPU only).

Follows the reference's one real test, /root/reference/tests/
test_rqs_accuracy.py:18-210: same three configurations, same parameter
generator shape, same four properties, same 1e-12 bounds in float64 -- run
against the C oracle and the independent NumPy restatement -- plus the checks
SURVEY.md 8(c) lists (identity at zero params, tails, nsf_symbol formula).
The reference test is property-only: absolute parity vs distrax is unpinned.
"""
import os

import numpy as np
import pytest

from oracle import numpy_flow as nf

# tests/test_rqs_accuracy.py:28-53
CONFIGS = [
  dict(num_bins=10, batch_size=100, num_features=2, range_min=-5.0, range_max=5.0, test_range=(-4.0, 4.0)),
  dict(num_bins=5, batch_size=50, num_features=1, range_min=-3.0, range_max=3.0, test_range=(-2.5, 2.5)),
  dict(num_bins=20, batch_size=200, num_features=3, range_min=-4.0, range_max=4.0, test_range=(-3.5, 3.5)),
]
MIN_SLOPE = 1e-3  # tests/test_rqs_accuracy.py:75


def gen_params(rng, batch, feat, K):
  """tests/test_rqs_accuracy.py:174-210 (uniform widths/heights normalised to
  sum 1, slopes U[0.5,2]); the spline re-normalises them through softmax."""
  w = rng.uniform(0.1, 2.0, (batch, feat, K)); w /= w.sum(-1, keepdims=True)
  h = rng.uniform(0.1, 2.0, (batch, feat, K)); h /= h.sum(-1, keepdims=True)
  s = rng.uniform(0.5, 2.0, (batch, feat, K + 1))
  return np.concatenate([w, h, s], axis=-1)


class COracleSpline:
  def __init__(self, oracle, theta, cfg):
    self.o, self.theta, self.cfg = oracle, theta, cfg
    self.K = cfg["num_bins"]

  def _run(self, v, inverse):
    th = np.broadcast_to(self.theta, v.shape + (3 * self.K + 1,)).reshape(-1, 3 * self.K + 1)
    out, ld = self.o.rqs(th, v.reshape(-1), self.K, self.cfg["range_min"],
                         self.cfg["range_max"], MIN_SLOPE, inverse=inverse)
    return out.reshape(v.shape), ld.reshape(v.shape)

  def forward(self, x): return self._run(x, False)
  def inverse(self, y): return self._run(y, True)


class NumpySpline:
  def __init__(self, theta, cfg):
    self.t = nf.rqs_tables(theta, cfg["range_min"], cfg["range_max"], 1e-4, MIN_SLOPE)

  def forward(self, x):
    t = [np.broadcast_to(a, x.shape + a.shape[-1:]) for a in self.t]
    return nf.rqs_forward(x, *t)

  def inverse(self, y):
    t = [np.broadcast_to(a, y.shape + a.shape[-1:]) for a in self.t]
    return nf.rqs_inverse(y, *t)


@pytest.fixture(params=["c", "numpy"])
def make_spline(request, oracle_lib):
  if request.param == "c":
    return lambda theta, cfg: COracleSpline(oracle_lib, theta, cfg)
  return lambda theta, cfg: NumpySpline(theta, cfg)


def test_rqs_reference_property_suite(make_spline):
  rng = np.random.default_rng(42)
  e_fi = e_if = e_jac = e_bnd = 0.0
  for cfg in CONFIGS:
    B, F, K = cfg["batch_size"], cfg["num_features"], cfg["num_bins"]
    theta = gen_params(rng, B, F, K)
    spl = make_spline(theta, cfg)
    lo, hi = cfg["test_range"]
    # 1: inverse(forward(x)) == x
    x = rng.uniform(lo, hi, (B, F))
    y, _ = spl.forward(x)
    xr, _ = spl.inverse(y)
    e_fi = max(e_fi, np.abs(xr - x).max())
    # 2: forward(inverse(y)) == y
    yt = rng.uniform(lo, hi, (B, F))
    xi, _ = spl.inverse(yt)
    yr, _ = spl.forward(xi)
    e_if = max(e_if, np.abs(yr - yt).max())
    # 3: sum logdet == log|det J| (J is diagonal: one scalar spline per feature).
    #    The reference differentiates with jax.jacfwd; here a complex-step-free
    #    high-order central difference in float64.
    if F <= 2:
      xj = rng.uniform(lo * 0.5, hi * 0.5, (F,))
      s1 = make_spline(theta[0], cfg)
      _, ld = s1.forward(xj)
      h = 1e-4
      f = lambda t: s1.forward(t)[0]
      deriv = (-f(xj + 2 * h) + 8 * f(xj + h) - 8 * f(xj - h) + f(xj - 2 * h)) / (12 * h)
      e_jac = max(e_jac, abs(ld.sum() - np.log(np.abs(deriv)).sum()))
    # 4: boundary behaviour (tests/test_rqs_accuracy.py:140-166)
    eps = 1e-6
    pts = np.array([[cfg["range_min"] + eps] * F, [cfg["range_max"] - eps] * F, [0.0] * F,
                    [lo * 0.5] * F, [hi * 0.5] * F])
    sb = make_spline(theta[:5], cfg)
    yb, _ = sb.forward(pts)
    xb, _ = sb.inverse(yb)
    e_bnd = max(e_bnd, np.abs(xb - pts).max())
  assert e_fi < 1e-12, e_fi
  assert e_if < 1e-12, e_if
  assert e_jac < 1e-9, e_jac   # 5-point FD: truncation-limited, not 1e-12
  assert e_bnd < 1e-12, e_bnd


def test_logdet_antisymmetry_and_c_numpy_agree(oracle_lib):
  rng = np.random.default_rng(7)
  for cfg in CONFIGS:
    B, F, K = cfg["batch_size"], cfg["num_features"], cfg["num_bins"]
    theta = gen_params(rng, B, F, K) * 3.0
    c, n = COracleSpline(oracle_lib, theta, cfg), NumpySpline(theta, cfg)
    x = rng.uniform(cfg["range_min"] - 2, cfg["range_max"] + 2, (B, F))  # includes tails
    yc, ldc = c.forward(x)
    yn, ldn = n.forward(x)
    assert np.abs(yc - yn).max() < 1e-13 and np.abs(ldc - ldn).max() < 1e-13
    xc, ildc = c.inverse(yc)
    xn, ildn = n.inverse(yn)
    assert np.abs(xc - xn).max() < 1e-13 and np.abs(ildc - ildn).max() < 1e-13
    assert np.abs(ldc + ildc).max() < 1e-12
    assert np.abs(xc - x).max() < 1e-12


def test_zero_params_is_identity(oracle_lib):
  """flows.py:48,71-76: zero `first` and zero-init output layer => theta = 0 =>
  equal bins on [-10,10] with slopes exactly 1 => the spline is the identity."""
  K = 5
  xk, yk, dl = oracle_lib.knots(np.zeros(3 * K + 1), K)
  assert np.array_equal(dl, np.ones(K + 1))
  assert np.allclose(xk, np.linspace(-10, 10, K + 1), atol=1e-14)
  assert np.array_equal(xk, yk)
  x = np.linspace(-13, 13, 1001)
  y, ld = oracle_lib.rqs(np.zeros((x.size, 3 * K + 1)), x, K, -10.0, 10.0, 1e-4)
  assert np.abs(y - x).max() < 1e-14
  assert np.abs(ld).max() < 1e-14


def test_tails_are_linear(oracle_lib):
  K = 5
  rng = np.random.default_rng(3)
  theta = rng.normal(size=3 * K + 1)
  xk, yk, dl = oracle_lib.knots(theta, K)
  x = np.array([-10.0, -10.5, -25.0, 10.0, 10.5, 40.0])
  y, ld = oracle_lib.rqs(np.tile(theta, (x.size, 1)), x, K, -10.0, 10.0, 1e-4)
  assert np.allclose(y[:3], (x[:3] + 10) * dl[0] - 10, atol=1e-13)
  assert np.allclose(y[3:], (x[3:] - 10) * dl[K] + 10, atol=1e-13)
  assert np.allclose(ld[:3], np.log(dl[0])) and np.allclose(ld[3:], np.log(dl[K]))


def test_forward_matches_reference_nsf_symbol(oracle_lib, golden_dir):
  """cnf_ot/models/nsf_symbol.py:6-13 is the one file of the reference's flow
  path that runs here; the fixture is its printed d f/d delta_k.  The oracle's
  forward spline must have exactly that sensitivity to the bin's left slope:
  dy/dtheta_slope[k] = (df/ddelta_k) * sigmoid(theta_slope[k] + offset)."""
  sympy = pytest.importorskip("sympy")
  with open(os.path.join(golden_dir, "nsf_symbol_dfddeltak.txt")) as f:
    expr = sympy.sympify(f.read().strip())
  names = "x xk xk1 yk yk1 deltak deltak1".split()
  fn = sympy.lambdify(sympy.symbols(names), expr, "numpy")
  K, m = 5, 1e-4
  rng = np.random.default_rng(11)
  offset = np.log(np.exp(1 - m) - 1)
  worst = 0.0
  for _ in range(50):
    theta = rng.normal(size=3 * K + 1)
    xk, yk, dl = oracle_lib.knots(theta, K)
    k = rng.integers(0, K)
    x = rng.uniform(xk[k], xk[k + 1])
    want = fn(x, xk[k], xk[k + 1], yk[k], yk[k + 1], dl[k], dl[k + 1])
    want *= 1.0 / (1.0 + np.exp(-(theta[2 * K + k] + offset)))
    h = 1e-5
    tp, tm = theta.copy(), theta.copy()
    tp[2 * K + k] += h; tm[2 * K + k] -= h
    yp, _ = oracle_lib.rqs(tp[None], [x], K, -10.0, 10.0, m)
    ym, _ = oracle_lib.rqs(tm[None], [x], K, -10.0, 10.0, m)
    worst = max(worst, abs((yp[0] - ym[0]) / (2 * h) - want))
  assert worst < 1e-8, worst


def test_f32_oracle_close_to_f64(oracle_lib):
  K = 5
  rng = np.random.default_rng(5)
  theta = rng.normal(0, 0.5, size=(2000, 3 * K + 1))
  x = rng.normal(size=2000) * 3
  y64, ld64 = oracle_lib.rqs(theta, x, K, -10.0, 10.0, 1e-4)
  y32, ld32 = oracle_lib.rqs(theta, x, K, -10.0, 10.0, 1e-4, dtype=np.float32)
  assert np.abs(y32 - y64).max() < 2e-5
  assert np.abs(ld32 - ld64).max() < 2e-5
