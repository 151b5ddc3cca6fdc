"""Vectorised float64 NumPy restatement of the flow path (TEST INFRASTRUCTURE).

A second, independently written oracle: where ``cnf_oracle_impl.h`` loops per
sample and picks the bin by index, this file follows the *array* form of the
published distrax algorithm (mask-and-sum bin selection, ``jnp.where`` tails),
so the two restatements check each other.  Parity status vs distrax itself:
**parity unpinned** (see ``cnf_oracle_impl.h``).

Reference lines restated:
  spline        distrax.RationalQuadraticSpline, call site flows.py:124-132
  conditioner   cnf_ot/models/flows.py:46-86
  coupling      cnf_ot/models/autoregressive.py:76-136
  composition   cnf_ot/models/conditional.py:159-177,233-243,316-321,382-402
  perms / base  cnf_ot/models/flows.py:141-143,166-173
"""
import numpy as np


def softplus(x):
  return np.logaddexp(x, 0.0)


def normalize_bin_sizes(u, total, min_bin):
  e = np.exp(u - u.max(axis=-1, keepdims=True))
  return e / e.sum(axis=-1, keepdims=True) * (total - u.shape[-1] * min_bin) + min_bin


def rqs_tables(theta, lo, hi, min_bin=1e-4, min_slope=1e-4, circular=False):
  """theta [...,3K+1] -> x_pos, y_pos, slopes, each [...,K+1].  circular: distrax
  boundary_slopes='circular' (flows.py:131): the last unnormalized slope := the first."""
  theta = np.asarray(theta, dtype=np.float64)
  K = (theta.shape[-1] - 1) // 3
  if circular:
    theta = np.concatenate([theta[..., :-1], theta[..., 2 * K:2 * K + 1]], -1)
  w = normalize_bin_sizes(theta[..., :K], hi - lo, min_bin)
  h = normalize_bin_sizes(theta[..., K:2 * K], hi - lo, min_bin)
  pad_lo = np.full(theta.shape[:-1] + (1,), lo)
  pad_hi = np.full(theta.shape[:-1] + (1,), hi)
  xk = np.concatenate([pad_lo, lo + np.cumsum(w[..., :-1], -1), pad_hi], -1)
  yk = np.concatenate([pad_lo, lo + np.cumsum(h[..., :-1], -1), pad_hi], -1)
  offset = np.log(np.exp(1.0 - min_slope) - 1.0)
  dl = softplus(theta[..., 2 * K:] + offset) + min_slope
  return xk, yk, dl


def _select(v, pos, xk, yk, dl):
  """Mask-and-sum 'gather' of the bin's two ends; no match -> first bin."""
  v = v[..., None]
  mask = (v >= pos[..., :-1]) & (v < pos[..., 1:])
  none = ~mask.any(axis=-1, keepdims=True)
  first = np.zeros_like(mask)
  first[..., 0] = True
  mask = np.where(none, first, mask).astype(np.float64)
  pick = lambda t: ((mask * t[..., :-1]).sum(-1), (mask * t[..., 1:]).sum(-1))
  return pick(xk), pick(yk), pick(dl)


def rqs_forward(x, xk, yk, dl):
  (x0, x1), (y0, y1), (d0, d1) = _select(x, xk, xk, yk, dl)
  bw, bh = x1 - x0, y1 - y0
  s = bh / bw
  z = np.clip((x - x0) / bw, 0.0, 1.0)
  sq_z, z1mz, sq_1mz = z * z, z - z * z, (1.0 - z) ** 2
  st = d1 + d0 - 2.0 * s
  den = s + st * z1mz
  y = y0 + bh * (s * sq_z + d0 * z1mz) / den
  ld = 2.0 * np.log(s) + np.log(d1 * sq_z + 2.0 * s * z1mz + d0 * sq_1mz) - 2.0 * np.log(den)
  below, above = x <= xk[..., 0], x >= xk[..., -1]
  y = np.where(below, (x - xk[..., 0]) * dl[..., 0] + yk[..., 0], y)
  y = np.where(above, (x - xk[..., -1]) * dl[..., -1] + yk[..., -1], y)
  ld = np.where(below, np.log(dl[..., 0]), ld)
  ld = np.where(above, np.log(dl[..., -1]), ld)
  return y, ld


def rqs_inverse(y, xk, yk, dl):
  (x0, x1), (y0, y1), (d0, d1) = _select(y, yk, xk, yk, dl)
  bw, bh = x1 - x0, y1 - y0
  s = bh / bw
  w = np.clip((y - y0) / bh, 0.0, 1.0)
  st = d1 + d0 - 2.0 * s
  c = -s * w
  b = d0 - st * w
  a = s - b
  z = np.clip(-2.0 * c / (b + np.sqrt(b * b - 4.0 * a * c)), 0.0, 1.0)
  x = bw * z + x0
  sq_z, z1mz, sq_1mz = z * z, z - z * z, (1.0 - z) ** 2
  den = s + st * z1mz
  ld = -2.0 * np.log(s) - np.log(d1 * sq_z + 2.0 * s * z1mz + d0 * sq_1mz) + 2.0 * np.log(den)
  below, above = y <= yk[..., 0], y >= yk[..., -1]
  x = np.where(below, (y - yk[..., 0]) / dl[..., 0] + xk[..., 0], x)
  x = np.where(above, (y - yk[..., -1]) / dl[..., -1] + xk[..., -1], x)
  ld = np.where(below, -np.log(dl[..., 0]), ld)
  ld = np.where(above, -np.log(dl[..., -1]), ld)
  return x, ld


class NumpyFlow:
  """The flow built from a flat parameter vector (layout: cnf_oracle_impl.h)."""

  def __init__(self, flat, D=2, L=2, H=16, M=2, K=5, lo=-10.0, hi=10.0,
               min_bin=1e-4, min_slope=1e-4, periodized=False):
    self.D, self.L, self.H, self.M, self.K = D, L, H, M, K
    self.periodized = periodized          # flows.py:58-64 (sin / cos features), :127-131 (range, circular slopes)
    self.lo, self.hi, self.min_bin, self.min_slope = lo, hi, min_bin, min_slope
    P = 3 * K + 1
    flat = np.asarray(flat, dtype=np.float64).reshape(-1)
    self.first = flat[:P]
    off = P
    self.cond = {}
    for l in range(L):
      for d in range(1, D):
        layers = []
        nin = (2 if periodized else 1) * (1 + d)
        for m in range(M):
          rows = nin if m == 0 else H
          W = flat[off:off + rows * H].reshape(rows, H); off += rows * H
          b = flat[off:off + H]; off += H
          layers.append((W, b))
        Wo = flat[off:off + H * P].reshape(H, P); off += H * P
        bo = flat[off:off + P]; off += P
        self.cond[(l, d)] = (layers, (Wo, bo))
    assert off == flat.size, (off, flat.size)

  def perm(self, l):
    p = np.arange(self.D)
    return p if l % 2 == 0 else p[::-1]

  def conditioner(self, l, d, inp):
    layers, (Wo, bo) = self.cond[(l, d)]
    h = inp
    for W, b in layers:
      h = np.maximum(h @ W + b, 0.0)
    return h @ Wo + bo

  def _tables(self, l, d, c, known, perm, B):
    if d == 0:
      theta = np.broadcast_to(self.first, (B, 3 * self.K + 1))
    else:
      inp = np.concatenate([c[:, None], known[:, perm[:d]]], axis=1)
      if self.periodized:
        inp = np.concatenate([np.sin(inp), np.cos(inp)], axis=1)
      theta = self.conditioner(l, d, inp)
    return rqs_tables(theta, self.lo, self.hi, self.min_bin, self.min_slope, circular=self.periodized)

  def _c(self, c, B):
    c = np.asarray(c, dtype=np.float64).reshape(-1)
    if c.size == 1:
      return np.full(B, c[0])
    assert B % c.size == 0
    return np.repeat(c, B // c.size)

  def forward_logdet(self, x, c):
    """base -> data (sampling direction; spline inverse; conditions on input)."""
    u = np.asarray(x, dtype=np.float64).copy()
    B = u.shape[0]
    c = self._c(c, B)
    total = np.zeros(B)
    for l in range(self.L):
      perm = self.perm(l)
      out = np.zeros_like(u)
      for d in range(self.D):
        i = perm[d]
        xk, yk, dl = self._tables(l, d, c, u, perm, B)
        out[:, i], ld = rqs_inverse(u[:, i], xk, yk, dl)
        total += ld
      u = out
    return u, total

  def inverse_logdet(self, y, c):
    """data -> base (log_prob direction; spline forward; conditions on output)."""
    u = np.asarray(y, dtype=np.float64).copy()
    B = u.shape[0]
    c = self._c(c, B)
    total = np.zeros(B)
    for l in reversed(range(self.L)):
      perm = self.perm(l)
      out = np.zeros_like(u)
      for d in range(self.D):
        i = perm[d]
        xk, yk, dl = self._tables(l, d, c, out, perm, B)
        out[:, i], ld = rqs_forward(u[:, i], xk, yk, dl)
        total += ld
      u = out
    return u, total

  @staticmethod
  def base_log_prob(x):
    return (-0.5 * x * x - 0.5 * np.log(2.0 * np.pi)).sum(-1)

  def log_prob(self, value, c):
    x, ildj = self.inverse_logdet(value, c)
    return self.base_log_prob(x) + ildj

  def sample_logprob(self, noise, c):
    noise = np.asarray(noise, dtype=np.float64)
    y, fldj = self.forward_logdet(noise, c)
    return y, self.base_log_prob(noise) - fldj
