"""float64 restatement of cnf_ot/mfc/applications.py and the two evaluators of
cnf_ot/utils.py over the oracle flow (TEST INFRASTRUCTURE ONLY).

Random draws are *inputs* here: the reference draws everything inside a loss
from one reused ``rng`` (applications.py:36-67,81-82,233-239), i.e. every
sampler call of a loss sees the same base noise, and the per-slice calls at
``batch_size // 32`` see the first rows of their own draw.  The build defines
noise as a pure function of (seed, element index), so "the draw of B//32
samples" is the first B//32 rows of the noise tensor handed in.  ``t_batch``
(applications.py:392,414,434) is also an input.
"""
import numpy as np

from . import capi


class OracleFlow:
  """model.apply.{sample,sample_and_log_prob,log_prob} on supplied noise."""

  def __init__(self, cfg: capi.OracleConfig, flat_params):
    self.cfg = cfg
    self.params = np.asarray(flat_params, dtype=np.float64).reshape(-1)

  def sample(self, noise, c):
    return capi.forward_logdet(self.cfg, self.params, noise, c)[0]

  def sample_and_log_prob(self, noise, c):
    return capi.sample_logprob(self.cfg, self.params, noise, c)

  def log_prob(self, value, c):
    return capi.log_prob(self.cfg, self.params, value, c)


MIXTURE_R = 5.0
MIXTURE_CENTERS = MIXTURE_R * np.array(  # applications.py:34-67
  [[0.0, 1.0], [1.0, 0.0], [0.0, -1.0], [-1.0, 0.0],
   [0.6, 0.8], [0.6, -0.8], [-0.6, -0.8], [-0.6, 0.8]])


def source_samples(kind, z, comp=None):
  """applications.py:23-71.  'mixture' is the live code (z + centre[comp], the
  same z for all eight components because the key is reused); 'gaussian' is
  the commented Gaussian source N(-3, .) of :28-32 that BASELINE's
  "Gaussian -> Gaussian" configs name: z @ cholesky(A) - 3."""
  z = np.asarray(z, dtype=np.float64)
  if kind == "mixture":
    assert z.shape[1] == 2, "the mixture source is 2-D (applications.py:40-67)"
    return z + MIXTURE_CENTERS[np.asarray(comp)]
  if kind == "gaussian":
    assert z.shape[1] == 2
    chol = np.linalg.cholesky(np.array([[5.0, 1.0], [1.0, 0.5]]))
    return z @ chol - 3.0
  raise ValueError(kind)


def kl_loss_fn(flow, T, cond, z, source="mixture", comp=None):
  """applications.py:11-86 (target N(0,I) drawn from the same key => same z)."""
  s1 = source_samples(source, z, comp)
  s2 = np.asarray(z, dtype=np.float64)
  samples = s1 * (T - cond) / T + s2 * cond / T
  return -flow.log_prob(samples, [cond]).mean()


def density_fit_kl_loss_fn(flow, T, z, source="mixture", comp=None):
  """applications.py:166-173"""
  return kl_loss_fn(flow, T, 0.0, z, source, comp) + kl_loss_fn(flow, T, T, z, source, comp)


def _mvn_iso_pdf(x, var):
  D = x.shape[1]
  return np.exp(-0.5 * (x * x).sum(1) / var) / np.sqrt((2.0 * np.pi * var) ** D)


def reverse_kl_loss_fn(flow, T, beta, cond, noise):
  """applications.py:129-163"""
  y, lp = flow.sample_and_log_prob(noise, [cond])
  src = _mvn_iso_pdf(y, 2.0 / beta * (T + 1.0))
  tgt = _mvn_iso_pdf(y, 2.0 / beta)
  return (lp - np.log(src * (T - cond) / T + tgt * cond / T)).mean()


def potential(subtype, r, a):
  """applications.py:181-193"""
  D = r.shape[1]
  if subtype == "quadratic":
    return (r ** 2).sum(1) / 2
  if subtype == "double_well":
    ones = a * np.ones((1, D))
    return (np.linalg.norm(r - ones, axis=1) * np.linalg.norm(r + ones, axis=1) / 2) ** 2
  if subtype == "obstacle":
    return 50 * np.exp(-(r ** 2).sum(1) / 2)
  raise ValueError(subtype)


def potential_loss_fn(flow, a, subtype, cond, noise):
  """applications.py:176-205"""
  y, _ = flow.sample_and_log_prob(noise, [cond])
  return potential(subtype, y, a).mean()


def kinetic_loss_fn(flow, dim, dt, cond, noise):
  """applications.py:220-242"""
  r1 = flow.sample(noise, [cond - dt / 2])
  r2 = flow.sample(noise, [cond + dt / 2])
  v = (r2 - r1) / dt
  return (v ** 2).mean() * dim / 2


def _score_fd(flow, r3, cond, dx):
  """central finite difference of log_prob, applications.py:264-273"""
  B, D = r3.shape
  score = np.zeros((B, D))
  for i in range(D):
    dr = np.zeros((1, D))
    dr[0, i] = dx / 2
    score[:, i] = (flow.log_prob(r3 + dr, [cond]) - flow.log_prob(r3 - dr, [cond])) / dx
  return score


def kinetic_with_score_loss_fn(flow, dim, beta, dt, dx, cond, noise):
  """applications.py:245-276"""
  r1 = flow.sample(noise, [cond - dt / 2])
  r2 = flow.sample(noise, [cond + dt / 2])
  r3 = flow.sample(noise, [cond])
  v = (r2 - r1) / dt + _score_fd(flow, r3, cond, dx) / beta
  return (v ** 2).mean() * dim / 2


def drift(subtype, r3, a):
  """Target drift of flow_matching_loss_fn.  'gradient' as the reference runs
  it is the 2-D "smiling" field that overwrites -a*r (applications.py:310,
  353-357); 'ou' is the documented OU drift -a*r (applications.py:310,
  README.md:58-61), the only form defined for dim != 2."""
  if subtype == "ou":
    return -a * r3
  if subtype == "gradient":
    assert r3.shape[1] == 2, "the reference's 'gradient' field is 2-D only"
    x, y = r3[:, 0], r3[:, 1]
    gx = -(x ** 2 + y ** 2 - 4) * x
    gy = -(x ** 2 + y ** 2 - 4) * y - 2 * (y - 1)
    return a * np.stack([gx, gy], axis=1)
  if subtype == "nongradient":
    assert r3.shape[1] == 2
    J = np.array([[0.0, 1.0], [-1.0, 0.0]])
    return -r3 * a + (r3 @ J) * 0.5
  if subtype == "lorenz":
    assert r3.shape[1] == 3
    _r = 9.0
    t = np.zeros_like(r3)
    t[:, 0] = 10 * (r3[:, 1] - r3[:, 0])
    t[:, 1] = _r * r3[:, 0] * (28 / _r - r3[:, 2]) - r3[:, 1]
    t[:, 2] = _r * r3[:, 0] * r3[:, 1] - r3[:, 2] * 8 / 3
    return t
  raise ValueError(subtype)


def flow_matching_loss_fn(flow, dim, a, sigma, subtype, cond, noise):
  """applications.py:279-374 (dt and dx are overridden to 0.01, :286,301)"""
  dt = dx = 0.01
  r1 = flow.sample(noise, [cond - dt / 2])
  r2 = flow.sample(noise, [cond + dt / 2])
  r3 = flow.sample(noise, [cond])
  v = (r2 - r1) / dt + _score_fd(flow, r3, cond, dx) * sigma
  return ((v - drift(subtype, r3, a)) ** 2).mean() * dim / 2


def ot_loss_fn(flow, dim, T, dt, subtype, _lambda, batch_size, noise, t_batch,
               source="mixture", comp=None):
  """applications.py:377-402; obstacle potential summed, not averaged (:397-400)"""
  loss = _lambda * density_fit_kl_loss_fn(flow, T, noise[:batch_size], source, comp)
  sub = noise[:batch_size // 32]
  for t in t_batch:
    loss += kinetic_loss_fn(flow, dim, dt, t, sub) / len(t_batch)
    if subtype == "obstacle":
      loss += potential_loss_fn(flow, 0, subtype, t, sub)
  return loss


def rwpo_loss_fn(flow, dim, T, beta, dt, dx, subtype, a, _lambda, batch_size,
                 noise, t_batch):
  """applications.py:405-421 (t_batch already scaled by T)"""
  loss = _lambda * reverse_kl_loss_fn(flow, T, beta, 0.0, noise[:batch_size]) + \
    potential_loss_fn(flow, a, subtype, T, noise[:batch_size])
  sub = noise[:batch_size // 32]
  for t in t_batch:
    loss += kinetic_with_score_loss_fn(flow, dim, beta, dt, dx, t, sub) / len(t_batch) * T
  return loss


def fp_loss_fn(flow, dim, T, a, sigma, subtype, _lambda, batch_size, noise,
               t_batch):
  """applications.py:424-441 (beta fixed to 4, :432)"""
  loss = _lambda * reverse_kl_loss_fn(flow, T, 4.0, 0.0, noise[:batch_size])
  sub = noise[:batch_size // 32]
  for t in t_batch:
    loss += flow_matching_loss_fn(flow, dim, a, sigma, subtype, t, sub) / len(t_batch) * T
  return loss


def calc_kinetic_energy(flow, dim, t_array, noise_per_slice):
  """cnf_ot/utils.py:311-340; noise_per_slice(k) -> [B,D] draw of slice k
  (the reference splits the key per slice, :328)."""
  dt = 0.01
  e_kin = 0.0
  for k, t in enumerate(t_array):
    z = noise_per_slice(k)
    r1 = flow.sample(z, [t - dt / 2])
    r2 = flow.sample(z, [t + dt / 2])
    e_kin += (((r2 - r1) / dt) ** 2).mean() / 2
  return e_kin / len(t_array) * dim


def calc_score_kinetic_energy(flow, dim, beta, t_array, noise_per_slice):
  """cnf_ot/utils.py:343-389"""
  dt = dx = 0.01
  e_kin = 0.0
  for k, t in enumerate(t_array):
    z = noise_per_slice(k)
    r1 = flow.sample(z, [t - dt / 2])
    r2 = flow.sample(z, [t + dt / 2])
    r3 = flow.sample(z, [t])
    v = (r2 - r1) / dt + _score_fd(flow, r3, t, dx) / beta
    e_kin += (v ** 2).mean() / 2
  return e_kin / len(t_array) * dim
