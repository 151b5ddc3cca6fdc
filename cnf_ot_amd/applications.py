"""Mirror of cnf_ot/mfc/applications.py on the fused MI355X loss kernels.

Same function names and positional arguments as the reference
(`ot_loss_fn(model, dim, T, dt, t_batch_size, subtype, params, rng, _lambda,
batch_size)` ...), so `functools.partial(applications.ot_loss_fn, model, dim, T,
dt, t_batch_size, subtype)` from cnf_ot/mfc/solvers.py:58-88 works unchanged.
Every term is ONE kernel launch that returns per-slice sums (no [B,D]
intermediate reaches HBM); a composite loss stacks the partial sums of all its
terms and does ONE sum all-reduce when samples are sharded over GPUs.

Random draws: the reference reuses one `rng` for every draw inside a loss
(applications.py:36-67,81-82,233-239,392).  Here `rng` is a seed for the
build's Philox stream: base noise is element (global sample index, dim) of
that stream -- so the `batch_size // 32` draw is the first rows of the full
draw and sharding does not change results -- and `t_batch` / mixture
components come from a host generator keyed by the same seed.

Reference quirks kept on purpose: `batch_size // 32`
per-slice batches; the obstacle potential is summed, not averaged, over slices
(applications.py:397-400); RWPO/FP kinetic scaled by T / t_batch_size;
flow_matching overrides dt = dx = 0.01 (:286,301); fp beta = 4 (:432).
"""
from typing import Optional, Sequence

import numpy as np
import torch

from . import _capi
from .distributed import Shard, all_reduce_sums, current_shard, shard_range
from .flows import DeviceRng, _OnDevice, _stream_ptr, seed_to_u64

# Cholesky factor L (lower: A = L L^T, applied as z @ L like oracle/losses.py) of the Gaussian source's covariance A = [[5, 1], [1, .5]]
# (applications.py:28-32), computed once: torch.linalg.cholesky of a 2 x 2 CPU tensor costs ~20 ms PER CALL on a
# 256-thread host (LAPACK thread start-up) -- it was 35 of the 41 ms of config 5's value_and_grad.
GAUSSIAN_SOURCE_CHOL = np.linalg.cholesky(np.array([[5.0, 1.0], [1.0, 0.5]])).astype(np.float32)
_CHOL_ROWS = {}        # device -> the factor's rows on that device
_CENTERS_DEV = {}      # device -> MIXTURE_CENTERS on that device
MIXTURE_R = 5.0
# applications.py:34-67: centres of the 8-mode mixture source
MIXTURE_CENTERS = MIXTURE_R * np.array(
  [[0.0, 1.0], [1.0, 0.0], [0.0, -1.0], [-1.0, 0.0],
   [0.6, 0.8], [0.6, -0.8], [-0.6, -0.8], [-0.6, 0.8]], dtype=np.float32)


def _spec(kind, subtype=0, dt=0.0, dx=0.0, coef=0.0, a=0.0, T=1.0, beta=1.0):
  return _capi.CnfLossSpec(kind, subtype, dt, dx, coef, a, T, beta)


def host_rng(rng, stream: int) -> np.random.Generator:
  """The host generator of a loss's non-Gaussian draws (time batch: stream 1, mixture components: stream 2), keyed by
  the loss's `rng`.  (One integer seed into PCG64: np.random.default_rng on a 4-word list costs 40 us per call -- a
  third of the host time of a config-3 loss evaluation.)"""
  seed, off = seed_to_u64(rng)
  return np.random.Generator(np.random.PCG64((seed << 72) | ((off & ((1 << 64) - 1)) << 8) | stream))


def draw_t_batch(rng, t_batch_size: int, scale: float = 1.0):
  """jax.random.uniform(rng, (t_batch_size,)) * T (applications.py:392,414,434): a host array, or -- for a DeviceRng --
  a device tensor drawn by cnf_fill_uniform_dev from the key in device memory."""
  if isinstance(rng, DeviceRng):
    out = torch.empty(t_batch_size, dtype=torch.float32, device=rng.device)
    with _OnDevice(rng.device):
      _capi.check(_capi.lib().cnf_fill_uniform_dev(rng.ptr, 0, t_batch_size, float(scale), out.data_ptr(),
                                                   _stream_ptr(rng.device)), "cnf_fill_uniform_dev")
    return out
  return (host_rng(rng, 1).uniform(0.0, 1.0, size=t_batch_size) * scale).astype(np.float32)


def _conds(t):
  """time-slice conditions as the term functions pass them on: a device tensor stays one, the rest a float32 array"""
  return t.reshape(-1) if torch.is_tensor(t) else np.atleast_1d(np.asarray(t, dtype=np.float32)).reshape(-1)


def _n_conds(t) -> int:
  return int(t.numel()) if torch.is_tensor(t) else int(np.atleast_1d(t).size)


def _cat_conds(parts):
  return torch.cat(list(parts)) if torch.is_tensor(parts[0]) else np.concatenate(list(parts))


def draw_components(rng, n: int) -> np.ndarray:
  """jax.random.choice(seed, a=8, shape=(n,)) (applications.py:36-38).  One byte per draw: the default int64 draw took
  5 ms per million on the GPU box's host and its [n, 2] table of centres another 5 plus a 16 MB copy -- an eager
  large-batch step with the mixture source was 15 ms of host work around 0.5 ms of kernels."""
  return host_rng(rng, 2).integers(0, 8, size=n, dtype=np.uint8)


class _Ctx:
  """Backend (loaded engine) + shard + local noise cache for one loss call.
  With `grad` (a zeroed flat float32 tensor) every term also accumulates
  coef * d(term sum)/d(params) into it (cnf_loss_terms_grad)."""

  def __init__(self, model, params, rng, shard: Optional[Shard], grad: Optional[torch.Tensor] = None):
    self.be = model.terms_backend(params)
    self.rng = rng
    self.shard = shard if shard is not None else current_shard()
    self.grad = grad
    self._noise = {}
    self._passes = []       # base -> data backward passes waiting for ONE launch (defer_pass_vjp)
    self._jobs = []         # fused loss terms waiting for ONE launch (cnf_loss_terms_grad_multi)
    self._grad64 = None     # the reduced gradient of the last `reduce`, in float64
    # the terms' per-slice sums land side by side in one buffer: `reduce` needs no concatenation kernel
    self._sumbuf = torch.empty(256, dtype=torch.float64, device=self.be.device) if hasattr(self.be, "device") and \
        getattr(self.be.device, "type", "cpu") == "cuda" else None
    self._sumpos = 0

  def new_sums(self, n: int):
    if self._sumbuf is None or self._sumpos + n > self._sumbuf.numel():
      return None
    v = self._sumbuf[self._sumpos:self._sumpos + n]
    self._sumpos += n
    return v

  def noise(self, n_global: int) -> torch.Tensor:
    """This rank's rows of the first n_global samples of the seed's stream."""
    if n_global not in self._noise:
      start, count = shard_range(n_global, self.shard)
      if self.shard.world == 1:      # one rank: a smaller draw is the first rows of a larger one already made
        for n_have, (z, _, _) in self._noise.items():
          if isinstance(n_have, int) and n_have >= n_global:
            self._noise[n_global] = (z[:n_global], 0, n_global)
            return self._noise[n_global]
      self._noise[n_global] = (self.be.normal(self.rng, count, first_sample=start), start, count)
    return self._noise[n_global]

  def terms(self, spec, pts, t, B_local, coef, shared=True):
    """per-slice sums of one term; `coef` = d(loss)/d(sum) (the same for every slice)."""
    t = _conds(t)
    kw = {}
    if self._sumbuf is not None:
      s = self.new_sums(_n_conds(t))
      if s is not None:
        kw["sums"] = s
    if self.grad is not None:
      if "sums" in kw and hasattr(self.be, "loss_terms_grad_multi"):
        # value_and_grad: the term is QUEUED -- the terms of a loss go out as one launch whose grid their tiles share
        # (flush_terms, at the latest in `reduce`); nobody reads a term's sums before the collective
        self._jobs.append((spec, pts, t, B_local, shared, coef, kw["sums"]))
        return kw["sums"]
      return self.be.loss_terms_grad(spec, pts, t, B_local, shared, coef, self.grad, **kw)
    return self.be.loss_terms(spec, pts, t, B_local, shared, **kw)

  def flush_terms(self):
    jobs, self._jobs = self._jobs, []
    if jobs:
      self.be.loss_terms_grad_multi(jobs, self.grad)

  def defer_pass_vjp(self, z, conds, count: int, ybar, ldbar):
    """Queue the backward of a base -> data pass (points z [S * count, D], one condition per slice of `count`
    points) instead of launching it: the passes of a loss's terms go out as ONE cnf_pass_vjp launch in `reduce`.
    A rank's share of a term is often too small to fill the GPU on its own (config 4: 512 and 1 536 waves for 2 048
    wave slots, each wave one tile -- two launches took two tile times, the merged one takes one)."""
    self._passes.append((z, _conds(conds), int(count), ybar, ldbar))

  def flush_passes(self):
    self.flush_terms()
    passes, self._passes = self._passes, []
    if not passes:
      return
    be = self.be
    if len(passes) == 1:
      z, conds, count, ybar, ldbar = passes[0]
      be.pass_vjp(z, be.slice_conds(conds), ybar, ldbar, False, grad=self.grad, want_xbar=False)
      return
    g = 0
    for _, _, count, _, _ in passes:
      g = int(np.gcd(g, count))
    dev = any(torch.is_tensor(c) for _, c, _, _, _ in passes)
    rep = []
    for _, c, count, _, _ in passes:
      if dev:
        # (the cached upload of slice_conds: a fresh host-to-device copy is not permitted inside a graph capture -- the
        # captured step failed from dim 6 up, where the reverse-KL pass [cond 0] is merged with the score term's)
        c = c if torch.is_tensor(c) else be.slice_conds(c)
        rep.append(c.repeat_interleave(count // g))
      else:
        rep.append(np.repeat(c, count // g))
    conds = _cat_conds(rep)
    z = torch.cat([p[0] for p in passes])
    ybar = torch.cat([p[3] if p[3] is not None else torch.zeros_like(p[0]) for p in passes])
    ldbar = None
    if any(p[4] is not None for p in passes):
      ldbar = torch.cat([p[4] if p[4] is not None else torch.zeros(p[0].shape[0], device=z.device) for p in passes])
    be.pass_vjp(z, be.slice_conds(conds), ybar, ldbar, False, grad=self.grad, want_xbar=False)

  def _stack(self, sums: Sequence[torch.Tensor]):
    """The partial sums of some terms as one float64 vector (a view of the sums buffer when they sit side by side)."""
    parts = [s.reshape(-1).to(torch.float64) for s in sums]
    n = sum(p.numel() for p in parts)
    if self._sumbuf is not None and len(parts) > 1:
      base = self._sumbuf.data_ptr()
      off = parts[0].data_ptr() - base
      ptr, ok = parts[0].data_ptr(), 0 <= off and off + 8 * n <= 8 * self._sumbuf.numel()
      for q in parts:
        ok = ok and q.data_ptr() == ptr
        ptr += 8 * q.numel()
      if ok:
        return self._sumbuf[off // 8:off // 8 + n], n
    return (torch.cat(parts) if len(parts) > 1 else parts[0]), n

  def reduce(self, sums: Sequence[torch.Tensor]) -> torch.Tensor:
    """The ONE collective of a loss evaluation: partial sums (+ the gradient)."""
    self.flush_passes()
    flat, n = self._stack(sums)
    if self.grad is not None and self.shard.world > 1:
      flat = all_reduce_sums(torch.cat([flat, self.grad.to(torch.float64)]), self.shard)
      self._grad64 = flat[n:]
      self.grad.copy_(flat[n:].to(self.grad.dtype))
      return flat[:n]
    return all_reduce_sums(flat, self.shard)

  # -- the collective of the FIRST terms under the compute of the later ones (config 5: BASELINE.json configs[4]) -----
  def reduce_begin(self, sums: Sequence[torch.Tensor]):
    """Start the sum all-reduce of the terms evaluated so far -- their partial sums and the gradient accumulated so
    far -- WITHOUT waiting for it (async_op: RCCL runs it on its own stream behind an event, gloo on a worker thread);
    the terms that follow accumulate into a fresh gradient buffer, and `reduce_end` folds the two.  With one rank
    nothing is started."""
    self.flush_passes()
    flat, n = self._stack(sums)
    work, g0 = None, self.grad
    if self.shard.world > 1:
      import torch.distributed as dist
      flat = torch.cat([flat, g0.to(torch.float64)]) if g0 is not None else flat.clone()
      work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.shard.group, async_op=True)
      if g0 is not None:
        self.grad = torch.zeros_like(g0)
    return flat, work, n, g0

  def reduce_end(self, handle) -> torch.Tensor:
    """Wait for `reduce_begin`'s collective (call it after the `reduce` of the later terms): the reduced sums of the
    first terms; the caller's gradient tensor receives first-phase + later-phase gradient."""
    flat, work, n, g0 = handle
    if work is not None:
      work.wait()
      if g0 is not None:      # first-phase + later-phase gradient, added in float64 and rounded once
        g0.copy_((flat[n:] + self._grad64).to(g0.dtype))
        self.grad = g0
    return flat[:n]

  def weighted(self, flat: torch.Tensor, sums: Sequence[torch.Tensor], coefs: Sequence[float]) -> torch.Tensor:
    """sum_i coefs[i] * (reduced) sums[i].sum(): one dot product with a cached weight vector instead of a slice /
    sum / multiply / add chain of tiny kernels per term (each costs a launch: 4-5 us on the stream)."""
    key = (tuple((float(c), int(s.numel())) for c, s in zip(coefs, sums)), str(flat.device))
    w = _WEIGHTS.get(key)
    if w is None:
      if len(_WEIGHTS) >= 64:
        _WEIGHTS.clear()
      w = _WEIGHTS[key] = torch.tensor(np.concatenate([np.full(n, c, dtype=np.float64) for c, n in key[0]]),
                                       dtype=torch.float64, device=flat.device)
    if flat.is_cuda:       # one block, fixed order, no BLAS call (and nothing a stream capture could not record)
      out = torch.empty(1, dtype=torch.float64, device=flat.device)
      with _OnDevice(flat.device):
        _capi.check(_capi.lib().cnf_weighted_sum(flat.data_ptr(), w.data_ptr(), flat.numel(), out.data_ptr(),
                                                 _stream_ptr(flat.device)), "cnf_weighted_sum")
      return out[0]
    return torch.dot(flat, w)

  def combine(self, sums: Sequence[torch.Tensor], coefs: Sequence[float]) -> torch.Tensor:
    """The loss from all its terms' partial sums, after the one collective."""
    return self.weighted(self.reduce(sums), sums, coefs)

  def combine_overlapped(self, first: Sequence[torch.Tensor], c_first: Sequence[float], later, c_later):
    """`combine` with the collective of the `first` terms already in flight while `later()` -- a callable that
    evaluates the remaining terms and returns their sums -- runs; a second, blocking collective for those."""
    h = self.reduce_begin(first)
    late = later()
    flat_late = self.reduce(late)
    flat_first = self.reduce_end(h)
    return self.weighted(torch.cat([flat_first, flat_late]), list(first) + list(late), list(c_first) + list(c_later))


# Sample-sharded value_and_grad of ot_loss_fn: start the all-reduce of the two density-fit terms (sums + their share
# of the gradient) as soon as they are done and run the t_batch_size kinetic / obstacle slices under it (BASELINE.json
# configs[4] "allreduce/compute overlap"); a second collective carries the rest.  False: ONE blocking collective at the
# end.  Both give the same loss and gradient (tests/test_distributed_cpu.py); see DESIGN.md 7 for what each costs.
OVERLAP_ALLREDUCE = True


_WEIGHTS = {}


# ---- local partial sums of each term ----------------------------------------

def _source_samples(ctx, z, start, count, n_global, source):
  if source == "mixture":      # applications.py:34-71 (live code)
    if z.shape[1] != 2:
      raise ValueError("the mixture source of kl_loss_fn is 2-D (applications.py:40-67)")
    if isinstance(ctx.rng, DeviceRng):
      out = torch.empty_like(z)
      with _OnDevice(z.device):
        _capi.check(_capi.lib().cnf_mixture_source_dev(ctx.rng.ptr, start, count, z.data_ptr(), out.data_ptr(), None,
                                                       _stream_ptr(z.device)), "cnf_mixture_source_dev")
      return out
    comp = draw_components(ctx.rng, n_global)[start:start + count]
    cdev = _CENTERS_DEV.get(z.device)
    if cdev is None:
      cdev = _CENTERS_DEV[z.device] = torch.from_numpy(MIXTURE_CENTERS).to(z.device)
    return z + cdev[torch.from_numpy(np.ascontiguousarray(comp)).to(z.device).long()]      # (the bytes cross, not the centres)
  if source == "gaussian":     # applications.py:28-32 (commented Gaussian source; BASELINE configs)
    if z.shape[1] != 2:
      raise ValueError("the Gaussian source N(-3, A) is 2-D (applications.py:28-32)")
    # z @ L - 3 for the 2 x 2 factor L, as three elementwise kernels: the rocBLAS GEMM torch picks for a [B, 2] x [2, 2]
    # product took 0.22 ms at B = 4.2 M (28 % of config 5's loss evaluation)
    rows = _CHOL_ROWS.get(z.device)
    if rows is None:
      rows = _CHOL_ROWS[z.device] = torch.from_numpy(np.concatenate([GAUSSIAN_SOURCE_CHOL, np.full((1, 2), -3.0, GAUSSIAN_SOURCE_CHOL.dtype)])).to(z.device)
    return torch.addcmul(torch.addcmul(rows[2], z[:, :1], rows[0]), z[:, 1:], rows[1])      # (two kernels, not three)
  raise ValueError(f"unknown source {source!r}")


# ---- dim 2, value_and_grad of large batches: the terms composed from table-path launches ---------------------------
# The fused gradient kernel evaluates the conditioner MLP per sample and multiplies per-sample weight gradients on
# the matrix cores (4 G flow passes/s); cnf_pass_vjp on the conditioner tables needs neither (per-piece sufficient
# statistics: DESIGN.md 5.4, 9 G passes/s).  So when a rank's share of a term is large enough for the tables
# (cnf_grad.hip: slices >= 8 192 points, >= 262 144 points per pass) the term is ONE forward launch on the tables,
# a few elementwise kernels for its value and adjoints, and ONE table backward launch.
TABLE_BACKWARD_MIN_SLICE = 8192
TABLE_BACKWARD_MIN_POINTS = 262144


def _use_table_backward(ctx, dim, count, n_slices, passes=1):
  be = ctx.be
  return (ctx.grad is not None and dim == 2 and hasattr(be, "pass_vjp") and getattr(be, "_tables_ok", False)
          and getattr(be, "_pwl_mode", 0) != 0 and count >= TABLE_BACKWARD_MIN_SLICE
          and count * n_slices * passes >= TABLE_BACKWARD_MIN_POINTS)


def _use_table_values(ctx, dim, count, n_slices, passes=1):
  """The same composition for the loss WITHOUT its gradient (table forward + term epilogue: 70 G flow passes/s where the
  fused loss kernel, three table sets in LDS, does 39 -- config 5's share 0.54 -> 0.3 ms)."""
  be = ctx.be
  return (ctx.grad is None and dim == 2 and hasattr(be, "kinetic_potential_vjp") and getattr(be, "_tables_ok", False)
          and getattr(be, "_pwl_mode", 0) != 0 and count >= TABLE_BACKWARD_MIN_SLICE
          and count * n_slices * passes >= TABLE_BACKWARD_MIN_POINTS)


def _neg_logprob_tables(ctx, samples, cond, coef):
  """-sum log_prob(samples; cond) and its gradient: data -> base pass, log_prob = base(x) + ildj."""
  be = ctx.be
  c = be.slice_conds([cond])
  # value and gradient from ONE launch over the data (cnf_neg_logprob_vjp: the backward kernel seeds itself)
  total = be.neg_logprob_vjp(samples, c, coef, ctx.grad) if ctx.grad is not None else None
  if total is not None:
    return total
  # plain fp32 positions, like the fused loss kernel: a mean over the batch does not need the float64 position path
  # that makes single log_prob values good to 1e-5 (1.75 x the time of this launch)
  was = getattr(be, "_precise", True)
  be.set_precise(False)
  try:
    x, ildj = be.inverse_logdet(samples, c)
  finally:
    be.set_precise(was)
  # d(-sum lp) scaled by coef: lp_bar = -coef; x_bar = lp_bar * d base/dx = coef * x; ld_bar = -coef
  total, xbar, ldbar = be.term_residual(_capi.TERM_NEG_LOGPROB, x, ildj, x.shape[0], loss_coef=coef,
                                        want_adjoints=ctx.grad is not None)
  if ctx.grad is not None:
    be.pass_vjp(samples, c, xbar, ldbar, True, grad=ctx.grad, want_xbar=False)
  return total


def _kinetic_tables(ctx, z, conds, count, dt, coef):
  """per-slice sums of |(r2 - r1) / dt|^2 with r1, r2 the same draw pushed to t -+ dt/2 (applications.py:220-242)."""
  be = ctx.be
  th = _conds(conds)
  S = _n_conds(th)
  half = np.float32(0.5 * dt)
  c2 = be.slice_conds(_cat_conds([th - half, th + half]))
  done = be.kinetic_potential_vjp(z, c2, S, dt, coef, ctx.grad) if hasattr(be, "kinetic_potential_vjp") else None
  if done is not None:      # (one call: no repeated copy of z, one table build, no adjoint scan)
    return done[0]
  z2 = z.repeat(2 * S, 1)
  r, _ = be.forward_logdet(z2, c2, want_logdet=False)
  sums, rbar, _ = be.term_residual(_capi.TERM_KINETIC, r, None, count, p0=dt, loss_coef=coef, want_adjoints=ctx.grad is not None)
  if ctx.grad is not None:
    be.pass_vjp(z2, c2, rbar, None, False, grad=ctx.grad, want_xbar=False)
  return sums


def _potential_tables(ctx, z, conds, count, subtype, a, coef):
  """per-slice sums of the potential at the samples pushed to the slices' times (applications.py:176-205)."""
  be = ctx.be
  th = _conds(conds)
  S = _n_conds(th)
  c = be.slice_conds(th)
  zr = z.repeat(S, 1) if S > 1 else z
  r, _ = be.forward_logdet(zr, c, want_logdet=False)
  sums, rbar, _ = be.term_residual(_capi.TERM_POTENTIAL, r, None, count, subtype=_capi.POTENTIALS[subtype], p0=a,
                                   loss_coef=coef, want_adjoints=ctx.grad is not None)
  if ctx.grad is not None:
    be.pass_vjp(zr, c, rbar, None, False, grad=ctx.grad, want_xbar=False)
  return sums


def _kinetic_potential_tables(ctx, z, conds, count, dt, c_kin, subtype, a, c_pot):
  """The kinetic and the potential term of ot_loss_fn's obstacle case (applications.py:392-400: the same draw pushed to
  t -+ dt/2 and to t) from ONE forward and ONE backward launch over the 3 S slices -- they were two of each."""
  be = ctx.be
  th = _conds(conds)
  S = _n_conds(th)
  n = S * count
  half = np.float32(0.5 * dt)
  c3 = be.slice_conds(_cat_conds([th - half, th + half, th]))
  done = (be.kinetic_potential_vjp(z, c3, S, dt, c_kin, ctx.grad, subtype=_capi.POTENTIALS[subtype], a=a, c_pot=c_pot)
          if hasattr(be, "kinetic_potential_vjp") else None)
  if done is not None:
    return done
  z3 = z.repeat(3 * S, 1)
  r, _ = be.forward_logdet(z3, c3, want_logdet=False)
  if ctx.grad is None:
    kin, _, _ = be.term_residual(_capi.TERM_KINETIC, r[:2 * n], None, count, p0=dt, loss_coef=c_kin, want_adjoints=False)
    pot, _, _ = be.term_residual(_capi.TERM_POTENTIAL, r[2 * n:], None, count, subtype=_capi.POTENTIALS[subtype], p0=a,
                                 loss_coef=c_pot, want_adjoints=False)
    return kin, pot
  rbar = torch.empty_like(r)
  kin, _, _ = be.term_residual(_capi.TERM_KINETIC, r[:2 * n], None, count, p0=dt, loss_coef=c_kin, rbar_out=rbar[:2 * n])
  pot, _, _ = be.term_residual(_capi.TERM_POTENTIAL, r[2 * n:], None, count, subtype=_capi.POTENTIALS[subtype], p0=a,
                               loss_coef=c_pot, rbar_out=rbar[2 * n:])
  be.pass_vjp(z3, c3, rbar, None, False, grad=ctx.grad, want_xbar=False)
  return kin, pot


def _kl_sum(ctx, T, cond, batch_size, source, coef):
  z, start, count = ctx.noise(batch_size)
  key = ("source", batch_size, source)       # the same key draws the same source samples for every condition
  s1 = ctx._noise.get(key)
  if s1 is None:
    s1 = ctx._noise[key] = _source_samples(ctx, z, start, count, batch_size, source)
  if cond == 0.0:
    samples = s1
  elif cond == T:
    samples = z
  else:
    samples = s1 * ((T - cond) / T) + z * (cond / T)      # target N(0,I) drawn from the same key
  if _use_table_backward(ctx, z.shape[1], count, 1) or _use_table_values(ctx, z.shape[1], count, 1):
    return _neg_logprob_tables(ctx, samples.contiguous(), cond, coef)
  return ctx.terms(_spec(_capi.TERM_NEG_LOGPROB), samples.contiguous(), [cond], count, coef)


def _reverse_kl_sum(ctx, T, beta, cond, batch_size, coef):
  z, _, count = ctx.noise(batch_size)
  if (_use_unfused(ctx, z.shape[1]) and count <= UNFUSED_RKL_MAX_BATCH) or _use_table_backward(ctx, z.shape[1], count, 1):
    return _reverse_kl_unfused(ctx, T, beta, cond, batch_size, coef)
  return ctx.terms(_spec(_capi.TERM_REVERSE_KL, T=T, beta=beta), z, [cond], count, coef)


def _potential_sum(ctx, a, subtype, conds, batch_size, coef):
  if subtype not in _capi.POTENTIALS:
    raise ValueError(f"unknown potential {subtype!r}")
  z, _, count = ctx.noise(batch_size)
  if _use_table_backward(ctx, z.shape[1], count, _n_conds(conds)) or _use_table_values(ctx, z.shape[1], count, _n_conds(conds)):
    return _potential_tables(ctx, z, conds, count, subtype, a, coef)
  return ctx.terms(_spec(_capi.TERM_POTENTIAL, subtype=_capi.POTENTIALS[subtype], a=a), z, conds, count, coef)


def _kinetic_sum(ctx, dt, conds, batch_size, coef):
  z, _, count = ctx.noise(batch_size)
  if (_use_table_backward(ctx, z.shape[1], count, _n_conds(conds), passes=2)
      or _use_table_values(ctx, z.shape[1], count, _n_conds(conds), passes=2)):
    return _kinetic_tables(ctx, z, conds, count, dt, coef)
  return ctx.terms(_spec(_capi.TERM_KINETIC, dt=dt), z, conds, count, coef)


# From this dimension on, the score terms run UNFUSED: the fused kernel runs a
# sample's 3 + 2*dim flow passes back to back on one lane, which starves the GPU
# when a rank has few samples (dim 10: 32 768 samples = 512 waves, 23 passes
# each).  Unfused, the 2*dim log_prob evaluations at r3 +- dx/2 e_d are just
# 2*dim*B more points for the fast flow kernel, and the gradient comes from
# torch.autograd over the differentiable flow passes (cnf_ot_amd.autograd).
UNFUSED_SCORE_MIN_DIM = 6
# ... and the reverse-KL term too while the rank's batch is small (the fused kernel runs one sample per lane:
# 131 072 samples are 2 waves per SIMD)
UNFUSED_RKL_MAX_BATCH = 131072


def _score_terms_unfused(ctx, conds, batch_size, dt, dx, coef_score, loss_coef, drift=-1, a=0.0):
  """per-slice sums of  sum_d ((r2-r1)/dt + coef_score * score_d(r3) - drift_d(r3))^2  from three launches:
  ONE base -> data pass over the 3 x S x count points (conditions t -+ dt/2 and t; at this size the
  wave-per-dimension kernel), ONE central-difference score launch over the S x count points r3 (its 2 D
  evaluation points per sample exist only inside the kernel) and ONE epilogue (cnf_score_residual).  With ctx.grad
  the epilogue also emits the adjoints, and two backward launches (cnf_logprob_fd_vjp, cnf_pass_vjp) accumulate
  the parameter gradient -- the reverse sweep written out by hand: no autograd graph, a dozen host calls."""
  be = ctx.be
  z, _, count = ctx.noise(batch_size)
  S = _n_conds(conds)
  n = S * count
  # conditions per SLICE (the engine broadcasts a slice's value over its `count` samples): 3 S floats, built on the
  # host and uploaded once per distinct time batch -- not per-sample tensors assembled by a chain of small kernels
  th = _conds(conds)
  half = np.float32(0.5 * dt)
  tt = be.slice_conds(th)
  c3 = be.slice_conds(_cat_conds([th - half, th + half, th]))
  z3 = z.repeat(3 * S, 1)                                     # the same draw for every slice and condition
  want = ctx.grad is not None
  r, _ = be.forward_logdet(z3, c3, want_logdet=False)
  if want and drift in (-1, _capi.DRIFTS["ou"]) and hasattr(be, "score_fd_vjp"):
    # value AND backward of the score term in one launch: the kernel that differentiates the 2 D evaluation points
    # forms the score from its own forward passes (no separate forward launch over them)
    sums, rbar = be.score_fd_vjp(r, tt, count, dt, dx, coef_score, drift, a, loss_coef, ctx.grad)
    ctx.defer_pass_vjp(z3, _cat_conds([th - half, th + half, th]), count, rbar, None)
    return sums
  r3 = r[2 * n:]
  score = be.logprob_fd(r3, tt, dx)
  sums, rbar, sbar = be.score_residual(r, score, count, dt, coef_score, drift, a, loss_coef, want)
  if want:
    r3bar = be.logprob_fd_vjp(r3, tt, dx, sbar, ctx.grad)     # adjoint of r3 through the score; + parameter gradient
    rbar[2 * n:] += r3bar
    ctx.defer_pass_vjp(z3, _cat_conds([th - half, th + half, th]), count, rbar, None)
  return sums


def _reverse_kl_unfused(ctx, T, beta, cond, batch_size, coef):
  """reverse_kl_loss_fn from one base -> data launch + an epilogue (+ one backward launch): the form for
  dimensions where a rank's share of the batch cannot fill the GPU from inside the fused kernel."""
  be = ctx.be
  z, _, count = ctx.noise(batch_size)
  c = be.slice_conds([cond])
  y, lp = be.sample_logprob(z, c)
  want = ctx.grad is not None
  total, ybar, lpbar = be.rkl_residual(y, lp, cond, T, beta, coef, want)
  if want:      # lp = base(noise) - fldj: the adjoint of the pass's log-det output is -lpbar
    if _use_table_backward(ctx, z.shape[1], count, 1):      # (dim 2, large batch: the table form, on its own)
      be.pass_vjp(z, c, ybar, -lpbar, False, grad=ctx.grad, want_xbar=False)
    else:
      ctx.defer_pass_vjp(z, [cond], count, ybar, -lpbar)
  return total


def _use_unfused(ctx, dim):
  return dim >= UNFUSED_SCORE_MIN_DIM and hasattr(ctx.be, "pass_vjp")


def _kinetic_score_sum(ctx, beta, dt, dx, conds, batch_size, coef):
  z, _, count = ctx.noise(batch_size)
  if _use_unfused(ctx, z.shape[1]):
    return _score_terms_unfused(ctx, conds, batch_size, dt, dx, 1.0 / beta, coef)
  return ctx.terms(_spec(_capi.TERM_KINETIC_SCORE, dt=dt, dx=dx, coef=1.0 / beta), z, conds, count, coef)


def _flow_matching_sum(ctx, dim, a, sigma, subtype, conds, batch_size, coef):
  if subtype not in _capi.DRIFTS:
    raise ValueError(f"unknown velocity field {subtype!r}")
  if subtype in ("nongradient",) and dim != 2:
    raise Exception("nongradient case is only implemented for 2D!")        # applications.py:359-360
  if subtype == "lorenz" and dim != 3:
    raise Exception("Lorenz dynamics is only defined for 3 dim!")          # applications.py:365-366
  if subtype == "gradient" and dim != 2:
    raise ValueError("the reference's 'gradient' target is a 2-D field (applications.py:353-357); "
                     "use subtype='ou' for the documented drift -a*r in other dimensions")
  z, _, count = ctx.noise(batch_size)
  if _use_unfused(ctx, dim) and subtype in ("ou", "lorenz"):
    return _score_terms_unfused(ctx, conds, batch_size, 0.01, 0.01, sigma, coef, _capi.DRIFTS[subtype], a)
  # dt and dx are overridden to 0.01 inside the reference function (:286,301)
  return ctx.terms(_spec(_capi.TERM_FLOW_MATCHING, subtype=_capi.DRIFTS[subtype], dt=0.01, dx=0.01,
                         coef=sigma, a=a), z, conds, count, coef)


# ---- public term functions (reference signatures) ----------------------------
# Every function takes two extra keywords: `shard` (cnf_ot_amd.distributed.Shard,
# default: the torch.distributed world) and `grad` (a zeroed flat float32 tensor
# that receives d loss / d params: see `value_and_grad`).

def kl_loss_fn(model, dim, T, params, cond, rng, batch_size, source="mixture", shard=None, grad=None):
  """applications.py:11-86: -mean log_prob of samples interpolated between the
  source and the target draw."""
  ctx = _Ctx(model, params, rng, shard, grad)
  return ctx.reduce([_kl_sum(ctx, T, float(cond), batch_size, source, 1.0 / batch_size)])[0] / batch_size


def density_fit_kl_loss_fn(model, dim, T, params, rng, batch_size, source="mixture", shard=None, grad=None):
  """applications.py:166-173"""
  ctx = _Ctx(model, params, rng, shard, grad)
  c = 1.0 / batch_size
  s = ctx.reduce([_kl_sum(ctx, T, 0.0, batch_size, source, c), _kl_sum(ctx, T, float(T), batch_size, source, c)])
  return (s[0] + s[1]) * c


def reverse_kl_loss_fn(model, dim, T, beta, params, cond, rng, batch_size, shard=None, grad=None):
  """applications.py:129-163"""
  ctx = _Ctx(model, params, rng, shard, grad)
  return ctx.reduce([_reverse_kl_sum(ctx, T, beta, float(cond), batch_size, 1.0 / batch_size)])[0] / batch_size


def potential_loss_fn(model, dim, a, subtype, params, cond, rng, batch_size, shard=None, grad=None):
  """applications.py:176-205"""
  ctx = _Ctx(model, params, rng, shard, grad)
  return ctx.reduce([_potential_sum(ctx, a, subtype, [float(cond)], batch_size, 1.0 / batch_size)])[0] / batch_size


def kinetic_loss_fn(model, dim, dt, params, cond, rng, batch_size, shard=None, grad=None):
  """applications.py:220-242: mean(v^2) * dim / 2 with v by finite differences in c."""
  ctx = _Ctx(model, params, rng, shard, grad)
  c = 0.5 / batch_size       # mean over batch*dim, times dim / 2
  return ctx.reduce([_kinetic_sum(ctx, dt, [float(cond)], batch_size, c)])[0] * c


def kinetic_with_score_loss_fn(model, dim, beta, dt, dx, params, cond, rng, batch_size, shard=None, grad=None):
  """applications.py:245-276"""
  ctx = _Ctx(model, params, rng, shard, grad)
  c = 0.5 / batch_size
  return ctx.reduce([_kinetic_score_sum(ctx, beta, dt, dx, [float(cond)], batch_size, c)])[0] * c


def flow_matching_loss_fn(model, dim, a, sigma, subtype, dt, dx, params, cond, rng, batch_size, shard=None,
                          grad=None):
  """applications.py:279-374 (dt, dx arguments are ignored, as in the reference)."""
  ctx = _Ctx(model, params, rng, shard, grad)
  c = 0.5 / batch_size
  return ctx.reduce([_flow_matching_sum(ctx, dim, a, sigma, subtype, [float(cond)], batch_size, c)])[0] * c


# ---- composite losses ---------------------------------------------------------

def ot_loss_fn(model, dim, T, dt, t_batch_size, subtype, params, rng, _lambda, batch_size,
               source="mixture", shard=None, grad=None, overlap=None):
  """applications.py:377-402.  overlap (sharded value_and_grad only; default OVERLAP_ALLREDUCE): the density-fit
  terms' all-reduce runs under the kinetic / obstacle slices."""
  ctx = _Ctx(model, params, rng, shard, grad)
  t_batch = draw_t_batch(rng, t_batch_size)
  sub = batch_size // 32
  c_kl, c_kin = _lambda / batch_size, 0.5 / (sub * t_batch_size)
  first = [_kl_sum(ctx, T, 0.0, batch_size, source, c_kl), _kl_sum(ctx, T, float(T), batch_size, source, c_kl)]

  def later():
    if subtype == "obstacle":      # summed, not averaged, over slices (applications.py:397-400)
      z, _, count = ctx.noise(sub)
      if (_use_table_backward(ctx, z.shape[1], count, _n_conds(t_batch), passes=2)      # one forward + one backward launch for both
          or _use_table_values(ctx, z.shape[1], count, _n_conds(t_batch), passes=2)):
        return list(_kinetic_potential_tables(ctx, z, t_batch, count, dt, c_kin, "obstacle", 0.0, 1.0 / sub))
      return [_kinetic_sum(ctx, dt, t_batch, sub, c_kin), _potential_sum(ctx, 0.0, "obstacle", t_batch, sub, 1.0 / sub)]
    return [_kinetic_sum(ctx, dt, t_batch, sub, c_kin)]

  c_later = [c_kin] + ([1.0 / sub] if subtype == "obstacle" else [])
  if (OVERLAP_ALLREDUCE if overlap is None else overlap) and ctx.shard.world > 1 and ctx.grad is not None:
    return ctx.combine_overlapped(first, [c_kl, c_kl], later, c_later)
  return ctx.combine(first + later(), [c_kl, c_kl] + c_later)


def rwpo_loss_fn(model, dim, T, beta, dt, dx, t_batch_size, subtype, a, params, rng, _lambda, batch_size,
                 shard=None, grad=None):
  """applications.py:405-421"""
  ctx = _Ctx(model, params, rng, shard, grad)
  t_batch = draw_t_batch(rng, t_batch_size, T)
  sub = batch_size // 32
  c_rkl, c_pot, c_kin = _lambda / batch_size, 1.0 / batch_size, 0.5 * T / (sub * t_batch_size)
  return ctx.combine([_reverse_kl_sum(ctx, T, beta, 0.0, batch_size, c_rkl),
                      _potential_sum(ctx, a, subtype, [float(T)], batch_size, c_pot),
                      _kinetic_score_sum(ctx, beta, dt, dx, t_batch, sub, c_kin)], [c_rkl, c_pot, c_kin])


def fp_loss_fn(model, dim, T, a, sigma, dt, dx, t_batch_size, subtype, params, rng, _lambda, batch_size,
               shard=None, grad=None):
  """applications.py:424-441 (beta = 4: the initial Gaussian has variance 1, :432)"""
  ctx = _Ctx(model, params, rng, shard, grad)
  t_batch = draw_t_batch(rng, t_batch_size, T)
  sub = batch_size // 32
  c_rkl, c_fm = _lambda / batch_size, 0.5 * T / (sub * t_batch_size)
  return ctx.combine([_reverse_kl_sum(ctx, T, 4.0, 0.0, batch_size, c_rkl),
                      _flow_matching_sum(ctx, dim, a, sigma, subtype, t_batch, sub, c_fm)], [c_rkl, c_fm])


def value_and_grad(loss_fn):
  """jax.value_and_grad(loss_fn) of cnf_ot/mfc/solvers.py:94 for the loss
  functions of this module (bound with functools.partial like the reference
  does): returns f(params, *args, **kw) -> (loss, grads) with `grads` a
  `Params` tree (haiku names) over one flat gradient tensor."""
  from .params import Params

  def wrapped(params, *args, **kw):
    if not isinstance(params, Params) or not params.flat.is_cuda:
      raise TypeError("value_and_grad needs device-resident cnf_ot_amd.Params")
    g = torch.zeros_like(params.flat)
    loss = loss_fn(params, *args, grad=g, **kw)
    return loss, Params(params.cfg, g)

  return wrapped
