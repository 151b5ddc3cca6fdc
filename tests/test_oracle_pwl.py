"""CPU check of the claim behind the dim-2 fast path: for a
fixed condition the reference's conditioner is a piecewise-linear function of
its scalar input with at most 289 pieces, and the per-piece affine maps the GPU
builds (cnf_ot_amd/csrc/cnf_pwl.h, restated in oracle/pwl_tables.py) reproduce
the network exactly.  The GPU tables themselves are checked end to end against
the oracle in tests/test_gpu_parity.py (-m gpu)."""
import numpy as np
import pytest

from oracle import pwl_tables as pt

N_W = 2 * 16 + 16 + 256 + 16 + 256 + 16        # one dim-2 conditioner, P = 16


def _zigzag(rng):
  w = np.zeros(N_W)
  w[16:32] = 1.0
  w[32:48] = -np.linspace(-7.5, 7.5, 16)
  slopes = np.array([1.0] + [-2.0, 2.0] * 7 + [-2.0])
  w[48:304] = (slopes[:, None] * (1.0 + 0.01 * np.arange(16))[None, :]).reshape(-1)
  w[304:320] = -0.5 + 0.02 * (np.arange(16) - 8)
  w[320:576] = rng.normal(0, 0.3, 256)
  w[576:592] = rng.normal(0, 0.3, 16)
  return w


@pytest.mark.parametrize("kind", ["small", "large", "zigzag", "dead_units", "zeros"])
def test_tables_reproduce_the_network(kind):
  rng = np.random.default_rng(7)
  for trial in range(6):
    if kind == "small":
      w = rng.normal(0, 0.2, N_W)
    elif kind == "large":
      w = rng.normal(0, 1.0, N_W)
    elif kind == "zigzag":
      w = _zigzag(rng)
    elif kind == "dead_units":          # first-layer units that ignore u (a_j = 0): no breakpoint of their own
      w = rng.normal(0, 0.5, N_W)
      w[16:32][rng.integers(0, 2, 16) == 0] = 0.0
    else:
      w = np.zeros(N_W)
    c = rng.uniform(0, 1)
    table = pt.build_table(w, c)
    bp = table[0]
    assert bp.size + 1 <= 289 and np.all(np.diff(bp) >= 0)
    if kind == "zigzag":
      assert bp.size > 200          # near the bound: what the GPU test uses to exercise rows past the LDS window
    u = np.concatenate([rng.normal(0, 4, 4000), rng.uniform(-40, 40, 1000), bp, bp + 1e-9, bp - 1e-9])
    want = pt.mlp(w, c, u)
    got = pt.eval_table(table, u)
    scale = 1.0 + np.abs(want).max()
    assert np.abs(got - want).max() <= 1e-10 * scale, (kind, trial, np.abs(got - want).max())


def test_piece_count_of_the_bench_parameters():
  """The parameter sets of bench.py / the tests have 30-50 pieces -- far inside the 128 rows kept in LDS."""
  rng = np.random.default_rng(42)
  flat = rng.normal(0.0, 0.2, 1200)
  for l in range(2):
    w = flat[16 + 592 * l: 16 + 592 * (l + 1)]
    for c in (0.0, 0.5, 1.0):
      n = pt.build_table(w, c)[0].size
      assert 8 <= n <= 128, n


@pytest.mark.parametrize("kind", ["small", "large", "zigzag", "dead_units"])
def test_weight_gradient_from_per_piece_statistics(kind):
  """on a piece of the conditioner tables both ReLU patterns are constant, so the weight
  gradient is linear in A = sum g and B = sum u g per piece.  oracle/pwl_grad.py restates that algebra; here it is
  checked against plain per-sample backprop through the network (and the input adjoint against the piece slopes)."""
  from oracle import pwl_grad as pg
  rng = np.random.default_rng(11)
  for trial in range(4):
    if kind == "small":
      w = rng.normal(0, 0.2, N_W)
    elif kind == "large":
      w = rng.normal(0, 1.0, N_W)
    elif kind == "zigzag":
      w = _zigzag(rng)
    else:
      w = rng.normal(0, 0.5, N_W)
      w[16:32][rng.integers(0, 2, 16) == 0] = 0.0
    c = float(rng.uniform(0, 1))
    u = rng.normal(0, 3.0, 4000)
    g = rng.normal(size=(4000, 16))
    table = pt.build_table(w, c)
    ref, du_ref = pg.grad_per_sample(w, c, u, g)
    A, B, piece = pg.piece_statistics(table, u, g)
    got = pg.grad_from_statistics(w, c, table, A, B)
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 1e-9 * scale, (kind, trial, np.abs(got - ref).max() / scale)
    du = pg.input_adjoint(table, u, g, piece)
    assert np.abs(du - du_ref).max() <= 1e-9 * np.abs(du_ref).max()
