"""Differentiable flow passes for torch.autograd.

`flow_forward(engine, flat_params, x, c)` / `flow_inverse(...)` return
(points, log|det J|) and are differentiable w.r.t. the parameters and the input
points: forward = the fused HIP flow kernel, backward = `cnf_pass_vjp` (the
backward kernels of cnf_grad.hip).  Any loss composed from these on the host
gets the gradients jax.value_and_grad gives the reference
(cnf_ot/mfc/solvers.py:94) -- the fused loss kernels of
`cnf_ot_amd.applications` are the fast path for the reference's own losses,
this is the general one (and the one that parallelises the 2*dim log_prob
evaluations of the score terms over the whole GPU at large dim).

The condition `c` is not differentiated (the reference differentiates w.r.t.
the parameters only; its time derivatives are finite differences).
"""
import torch

from . import _capi
from .flows import FlowEngine, _stream_ptr


class _FlowPass(torch.autograd.Function):

  @staticmethod
  def forward(ctx, flat, x, c, engine: FlowEngine, to_base: bool):
    engine.load(flat)
    x = x.contiguous()
    out, ld = (engine.inverse_logdet if to_base else engine.forward_logdet)(x, c)
    ctx.save_for_backward(flat, x, c)
    ctx.engine, ctx.to_base = engine, to_base
    return out, ld

  @staticmethod
  def backward(ctx, g_out, g_ld):
    flat, x, c = ctx.saved_tensors
    eng = ctx.engine
    eng.load(flat)
    need_p, need_x = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    grad = torch.zeros_like(flat) if need_p else None
    xbar = eng.pass_vjp(x, c, g_out, g_ld, ctx.to_base, grad=grad, want_xbar=need_x)
    return grad, xbar, None, None, None


def flow_forward(engine: FlowEngine, flat_params: torch.Tensor, x: torch.Tensor, c):
  """base -> data: (y, log|det J|), differentiable in flat_params and x."""
  return _FlowPass.apply(flat_params, x, _cond_tensor(engine, c), engine, False)


def flow_inverse(engine: FlowEngine, flat_params: torch.Tensor, y: torch.Tensor, c):
  """data -> base: (x, log|det J^-1|), differentiable in flat_params and y."""
  return _FlowPass.apply(flat_params, y, _cond_tensor(engine, c), engine, True)


def log_prob(engine: FlowEngine, flat_params: torch.Tensor, value: torch.Tensor, c):
  """ConditionalTransformed.log_prob (conditional.py:316-321), differentiable."""
  x, ildj = flow_inverse(engine, flat_params, value, c)
  return (-0.5 * x * x).sum(1) - 0.5 * x.shape[1] * 1.8378770664093453 + ildj


class _LogProbFD(torch.autograd.Function):
  """score[i, d] = (log_prob(r_i + dx/2 e_d) - log_prob(r_i - dx/2 e_d)) / dx (applications.py:264-273):
  forward cnf_logprob_fd, backward cnf_logprob_fd_vjp -- the 2 D evaluation points per sample exist only inside
  the kernels."""

  @staticmethod
  def forward(ctx, flat, pts, c, engine: FlowEngine, dx: float):
    engine.load(flat)
    pts = pts.contiguous()
    score = engine.logprob_fd(pts, c, dx)
    ctx.save_for_backward(flat, pts, c)
    ctx.engine, ctx.dx = engine, dx
    return score

  @staticmethod
  def backward(ctx, gbar):
    flat, pts, c = ctx.saved_tensors
    eng = ctx.engine
    eng.load(flat)
    grad = torch.zeros_like(flat)
    pts_bar = eng.logprob_fd_vjp(pts, c, ctx.dx, gbar.contiguous(), grad, want_pts_bar=ctx.needs_input_grad[1])
    return (grad if ctx.needs_input_grad[0] else None), pts_bar, None, None, None


def logprob_fd(engine: FlowEngine, flat_params: torch.Tensor, pts: torch.Tensor, c, dx: float):
  """Central-difference score of log_prob, differentiable in flat_params and pts."""
  return _LogProbFD.apply(flat_params, pts, _cond_tensor(engine, c), engine, float(dx))


def _cond_tensor(engine, c):
  if not torch.is_tensor(c):
    c = torch.as_tensor(c, dtype=torch.float32)
  return c.to(device=engine.device, dtype=torch.float32).reshape(-1)
