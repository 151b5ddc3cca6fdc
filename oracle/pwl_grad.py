"""TEST INFRASTRUCTURE -- the weight gradient of the dim-2 conditioner from PER-PIECE SUFFICIENT STATISTICS
("next": the table form of the backward pass), restated in float64 NumPy and checked against
the per-sample backward of the network itself (tests/test_oracle_pwl.py).

For a fixed condition c the conditioner theta(u) = Wo^T relu(W1^T relu(a u + b) + b1) + bo is piecewise linear in
its scalar input u (oracle/pwl_tables.py).  On one piece both ReLU activity patterns m1, m2 are constant and
h1 = m1 * (a u + b), h2 = m2 * (P u + Q), theta are affine in u, so for output adjoints g_i = d loss / d theta(u_i)
every parameter gradient is LINEAR in two 16-vectors per piece,

    A = sum_i g_i          B = sum_i u_i g_i        (i over the samples whose u_i lies in the piece):

    d bo  += A                                  d Wo += P (x) B + Q (x) A        (P, Q zero where m2 is off)
    G2  = m2 * (Wo A),  G2u = m2 * (Wo B)       d b1 += G2      d W1 += (m1 a) (x) G2u + (m1 b) (x) G2
    G1  = m1 * (W1 G2), G1u = m1 * (W1 G2u)     d b0 += G1      d W0[c row] += c G1      d W0[u row] += G1u

and the adjoint of u_i itself is S . g_i with the piece's slope vector S.  A flow kernel on the tables therefore
needs, per sample, only the spline partials (theta adjoints), 16 FMAs for the input adjoint and 32 accumulations;
the products above run once per piece.

Never imported by the product (cnf_ot_amd/)."""
import numpy as np

from .pwl_tables import H, split_conditioner, build_table, _test_point


def grad_per_sample(w, c, u, g):
  """Reference: plain backprop through the network, sample by sample (vectorised).  u [N], g [N, P] ->
  (flat gradient in the layout of `w`, du [N])."""
  W0, b0, W1, b1, Wo, bo = split_conditioner(w)
  x = np.stack([np.full_like(u, c), u], axis=1)
  z1 = x @ W0 + b0; h1 = np.maximum(z1, 0.0)
  z2 = h1 @ W1 + b1; h2 = np.maximum(z2, 0.0)
  dbo = g.sum(0); dWo = h2.T @ g
  g2 = (g @ Wo.T) * (z2 > 0.0)
  db1 = g2.sum(0); dW1 = h1.T @ g2
  g1 = (g2 @ W1.T) * (z1 > 0.0)
  db0 = g1.sum(0); dW0 = x.T @ g1
  du = g1 @ W0[1]
  return np.concatenate([dW0.ravel(), db0, dW1.ravel(), db1, dWo.ravel(), dbo]), du


def piece_statistics(table, u, g):
  """A [n+1, P], B [n+1, P]: sums of g and of u g over the samples of every piece."""
  bp = table[0]
  p = np.searchsorted(bp, u, side="left")
  n_pieces = bp.size + 1
  A = np.zeros((n_pieces, g.shape[1])); B = np.zeros_like(A)
  np.add.at(A, p, g)
  np.add.at(B, p, g * u[:, None])
  return A, B, p


def grad_from_statistics(w, c, table, A, B):
  """The flat gradient from the per-piece statistics alone (no sample is looked at)."""
  W0, b0, W1, b1, Wo, bo = split_conditioner(w)
  a, b = W0[1], W0[0] * c + b0
  bp = table[0]
  n = bp.size
  dW0 = np.zeros_like(W0); db0 = np.zeros(H); dW1 = np.zeros_like(W1); db1 = np.zeros(H)
  dWo = np.zeros_like(Wo); dbo = np.zeros(Wo.shape[1])
  for p in range(n + 1):
    if not (A[p].any() or B[p].any()):
      continue
    lo = -np.inf if p == 0 else bp[p - 1]
    hi = bp[p] if p < n else np.inf
    ut = _test_point(lo, hi)                       # any interior point fixes the activity patterns of the piece
    m1 = (a * ut + b > 0.0).astype(np.float64)
    P = (W1 * (m1 * a)[:, None]).sum(0)
    Q = (W1 * (m1 * b)[:, None]).sum(0) + b1
    m2 = (P * ut + Q > 0.0).astype(np.float64)
    P, Q = P * m2, Q * m2
    dbo += A[p]
    dWo += np.outer(P, B[p]) + np.outer(Q, A[p])
    G2, G2u = m2 * (Wo @ A[p]), m2 * (Wo @ B[p])
    db1 += G2
    dW1 += np.outer(m1 * a, G2u) + np.outer(m1 * b, G2)
    G1, G1u = m1 * (W1 @ G2), m1 * (W1 @ G2u)
    db0 += G1
    dW0[0] += c * G1
    dW0[1] += G1u
  return np.concatenate([dW0.ravel(), db0, dW1.ravel(), db1, dWo.ravel(), dbo])


def input_adjoint(table, u, g, p):
  """du_i = S[piece of u_i] . g_i"""
  return (table[1][p] * g).sum(1)
