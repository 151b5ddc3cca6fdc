"""TEST INFRASTRUCTURE -- float64 NumPy restatement of the conditioner tables
that cnf_ot_amd/csrc/cnf_pwl.h builds on the GPU (pwl_build_kernel), used to
check the CLAIM behind the dim-2 fast path on the CPU: for a fixed condition c
the reference's conditioner (hk.nets.MLP [16, 16] + zero-init hk.Linear,
cnf_ot/models/flows.py:146-158, input [c, u]) is a continuous piecewise-linear
function of the scalar u with at most 1 + 16 + 17*16 = 289 pieces, and the
per-piece affine maps reproduce it exactly.

Never imported by the product (cnf_ot_amd/)."""
import numpy as np

H = 16


def split_conditioner(w):
  """Flat layout of one dim-2 conditioner: W0[2][16] | b0[16] | W1[16][16] | b1[16] | Wout[16][P] | bout[P]."""
  w = np.asarray(w, dtype=np.float64)
  P = (w.size - (2 * H + H + H * H + H)) // (H + 1)
  o = 0
  W0 = w[o:o + 2 * H].reshape(2, H); o += 2 * H
  b0 = w[o:o + H]; o += H
  W1 = w[o:o + H * H].reshape(H, H); o += H * H
  b1 = w[o:o + H]; o += H
  Wo = w[o:o + H * P].reshape(H, P); o += H * P
  bo = w[o:o + P]
  return W0, b0, W1, b1, Wo, bo


def mlp(w, c, u):
  """theta(u) of the network itself, u: [N] -> [N, P]."""
  W0, b0, W1, b1, Wo, bo = split_conditioner(w)
  x = np.stack([np.full_like(u, c), u], axis=1)
  h1 = np.maximum(x @ W0 + b0, 0.0)
  h2 = np.maximum(h1 @ W1 + b1, 0.0)
  return h2 @ Wo + bo


def _test_point(lo, hi):
  fl, fh = np.isfinite(lo), np.isfinite(hi)
  return 0.5 * (lo + hi) if fl and fh else (lo + 1.0 if fl else (hi - 1.0 if fh else 0.0))


def build_table(w, c):
  """(breakpoints [n] sorted, S [n+1, P], T [n+1, P], u_ref [n+1]): piece p covers
  (bp[p-1], bp[p]] and theta(u) = S[p] (u - u_ref[p]) + T[p] on it."""
  W0, b0, W1, b1, Wo, bo = split_conditioner(w)
  a, b = W0[1], W0[0] * c + b0
  with np.errstate(divide="ignore", invalid="ignore"):
    bp1 = np.sort(np.where(a != 0.0, -b / a, np.inf))
  cands = list(bp1[np.isfinite(bp1)])
  edges = np.concatenate([[-np.inf], bp1, [np.inf]])
  for i in range(H + 1):
    lo, hi = edges[i], edges[i + 1]
    if not (lo < hi and lo < np.inf):
      continue
    u = _test_point(lo, hi)
    on = (a * u + b > 0.0)
    Pk = (W1 * (on * a)[:, None]).sum(0)
    Qk = (W1 * (on * b)[:, None]).sum(0) + b1
    with np.errstate(divide="ignore", invalid="ignore"):
      r = np.where(Pk != 0.0, -Qk / Pk, np.inf)
    cands += list(r[(r > lo) & (r < hi)])
  bp = np.sort(np.asarray(cands, dtype=np.float64))
  n = bp.size
  S, T, ref = [], [], []
  for p in range(n + 1):
    lo = -np.inf if p == 0 else bp[p - 1]
    hi = bp[p] if p < n else np.inf
    u = _test_point(lo, hi)
    on1 = (a * u + b > 0.0)
    Pk = (W1 * (on1 * a)[:, None]).sum(0)
    Qk = (W1 * (on1 * b)[:, None]).sum(0) + b1
    on2 = (Pk * u + Qk > 0.0)
    s = (Wo * (on2 * Pk)[:, None]).sum(0)
    t = (Wo * (on2 * Qk)[:, None]).sum(0) + bo
    S.append(s); T.append(t + s * u); ref.append(u)
  return bp, np.asarray(S), np.asarray(T), np.asarray(ref)


def eval_table(table, u):
  bp, S, T, ref = table
  p = np.searchsorted(bp, u, side="left")        # pieces are (bp[p-1], bp[p]]
  return S[p] * (u - ref[p])[:, None] + T[p]
