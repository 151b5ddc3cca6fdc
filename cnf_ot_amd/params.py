"""Parameter container of the flow: the haiku parameter tree of the reference
as views into ONE flat float32 tensor (the layout of the C ABI).

Reference naming (cnf_ot/models/flows.py:46-86,146-158):
  '~' / 'first'                                   shape (1, P)   zero-init
  'mlp_layer{l}_d{d}/~/linear_{m}' / 'w','b'      hk.nets.MLP hidden layers
  'linear_out_layer{l}_d{d}' / 'w','b'            hk.Linear(P), zero-init
with P = 3*num_bins + 1 and d = 1..D-1 (the d=0 spline of every layer shares
'first').  The reference has no on-disk format (params live only in the Python
process, solvers.py:54); `save_npz` / `load_npz` define one keyed by these
names so JAX-trained parameters can be exported with
``np.savez(path, **{f"{mod}/{name}": arr for mod, d in params.items() for name, arr in d.items()})``.
"""
from dataclasses import dataclass
from functools import lru_cache
from typing import Dict, Iterator, List, Tuple

import numpy as np
import torch


@dataclass(frozen=True)
class FlowConfig:
  """config/mfc.yaml:29-33 + general.dim, and the spline constants that
  flows.py:124-132 passes to distrax.RationalQuadraticSpline."""
  dim: int = 2
  num_layers: int = 2
  hidden_size: int = 16
  mlp_num_layers: int = 2
  num_bins: int = 5
  range_min: float = -10.0
  range_max: float = 10.0
  min_bin_size: float = 1e-4
  min_knot_slope: float = 1e-4
  # flows.py:58-64,127-131: the conditioner MLP sees [sin(x), cos(x)] of its input x = [c, v], the splines live on
  # [0, 2 pi] (FlowConfig.torus sets the range) with boundary_slopes='circular'.  Flow functions only: the loss and
  # gradient kernels do not take it (no reference call site does either).
  periodized: bool = False

  @staticmethod
  def torus(**kw) -> "FlowConfig":
    import math
    return FlowConfig(range_min=0.0, range_max=2.0 * math.pi, periodized=True, **kw)

  @property
  def num_bijector_params(self) -> int:
    return 3 * self.num_bins + 1

  def param_count(self) -> int:
    return _param_count(self)


@lru_cache(maxsize=None)
def _param_count(cfg: "FlowConfig") -> int:
  return sum(n for _, _, _, n in _spec_sizes(cfg))


@lru_cache(maxsize=None)
def _spec_sizes(cfg: "FlowConfig"):
  return tuple((m, n, s, int(np.prod(s))) for m, n, s in param_spec(cfg))


@lru_cache(maxsize=None)
def param_spec(cfg: FlowConfig) -> List[Tuple[str, str, Tuple[int, ...]]]:
  """(module, name, shape) in flat-layout order."""
  H, P = cfg.hidden_size, cfg.num_bijector_params
  spec = [("~", "first", (1, P))]
  for l in range(cfg.num_layers):
    for d in range(1, cfg.dim):
      name = f"layer{l}_d{d}"
      for m in range(cfg.mlp_num_layers):
        rows = ((2 if cfg.periodized else 1) * (1 + d)) if m == 0 else H
        spec.append((f"mlp_{name}/~/linear_{m}", "w", (rows, H)))
        spec.append((f"mlp_{name}/~/linear_{m}", "b", (H,)))
      spec.append((f"linear_out_{name}", "w", (H, P)))
      spec.append((f"linear_out_{name}", "b", (P,)))
  return spec


class Params(dict):
  """``{module: {name: tensor}}`` whose leaves are views into ``self.flat``.
  Behaves like the reference's immutable nested dict; in-place updates of
  ``flat`` (an optimiser step) are seen by every leaf."""

  def __init__(self, cfg: FlowConfig, flat: torch.Tensor):
    super().__init__()
    if flat.dtype != torch.float32 or flat.dim() != 1 or not flat.is_contiguous():
      raise ValueError("flat parameter tensor must be contiguous 1-D float32")
    if flat.numel() != cfg.param_count():
      raise ValueError(f"flat has {flat.numel()} values, config needs {cfg.param_count()}")
    self.cfg = cfg
    self.flat = flat
    self._built = False        # the tree of views is built on first use: an optimiser step only needs `flat`

  def _build(self):
    if not self._built:
      self._built = True
      off = 0
      for mod, name, shape, n in _spec_sizes(self.cfg):
        dict.setdefault(self, mod, {})[name] = self.flat[off:off + n].view(*shape)
        off += n
    return self

  def __getitem__(self, k): return dict.__getitem__(self._build(), k)
  def __iter__(self): return dict.__iter__(self._build())
  def __len__(self): return dict.__len__(self._build())
  def __contains__(self, k): return dict.__contains__(self._build(), k)
  def keys(self): return dict.keys(self._build())
  def items(self): return dict.items(self._build())
  def values(self): return dict.values(self._build())
  def get(self, k, default=None): return dict.get(self._build(), k, default)
  def __repr__(self): return dict.__repr__(self._build())
  def __eq__(self, other): return dict.__eq__(self._build(), other)
  __hash__ = None

  @classmethod
  def zeros(cls, cfg: FlowConfig, device="cpu") -> "Params":
    """init_flow_to_identity=True (flows.py:48,71-76): `first` and the output
    layers are zero; the hidden layers are not zero in the reference (haiku's
    truncated-normal default) -- see `init`."""
    return cls(cfg, torch.zeros(cfg.param_count(), dtype=torch.float32, device=device))

  @classmethod
  def init(cls, cfg: FlowConfig, seed: int = 0, device="cpu") -> "Params":
    """model.init (solvers.py:54): hk.Linear default init for hidden layers --
    w ~ TruncatedNormal(stddev=1/sqrt(fan_in)), b = 0 -- and zeros for `first`
    and linear_out (identity flow).  haiku's exact stream is JAX threefry and
    is not reproduced; the distribution is."""
    p = cls.zeros(cfg, "cpu")
    gen = torch.Generator().manual_seed(int(seed))
    for mod, leaves in p.items():
      if mod.startswith("mlp_"):
        w = leaves["w"]
        std = 1.0 / np.sqrt(w.shape[0])
        torch.nn.init.trunc_normal_(w, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=gen)
    return p.to(device)

  @classmethod
  def random(cls, cfg: FlowConfig, scale: float, seed: int = 0, device="cpu") -> "Params":
    """N(0, scale^2) on every tensor incl. `first` (numpy PCG64 stream, the
    synthetic parameter sets of tests/ and bench.py)."""
    rng = np.random.default_rng(seed)
    flat = torch.from_numpy(rng.normal(0.0, scale, cfg.param_count()).astype(np.float32))
    return cls(cfg, flat.to(device))

  def to(self, device) -> "Params":
    return Params(self.cfg, self.flat.to(device).contiguous())

  def clone(self) -> "Params":
    return Params(self.cfg, self.flat.clone())

  def leaves(self) -> Iterator[Tuple[str, str, torch.Tensor]]:
    for mod, name, _ in param_spec(self.cfg):
      yield mod, name, self[mod][name]

  def save_npz(self, path: str) -> None:
    np.savez(path, **{f"{mod}/{name}": t.detach().cpu().numpy() for mod, name, t in self.leaves()})

  @classmethod
  def load_npz(cls, cfg: FlowConfig, path: str, device="cpu") -> "Params":
    with np.load(path, allow_pickle=False) as z:
      return from_tree(cfg, {k: z[k] for k in z.files}, device)


def from_tree(cfg: FlowConfig, tree: Dict, device="cpu") -> Params:
  """Accepts a nested ``{module: {name: array}}`` (haiku form) or a flat
  ``{"module/name": array}`` mapping of numpy arrays / tensors (float64
  reference parameters are rounded to float32 here)."""
  chunks = []
  for mod, name, shape in param_spec(cfg):
    if mod in tree and isinstance(tree[mod], dict):
      leaf = tree[mod][name]
    else:
      leaf = tree[f"{mod}/{name}"]
    t = torch.as_tensor(np.asarray(leaf.detach().cpu() if torch.is_tensor(leaf) else leaf))
    if tuple(t.shape) != tuple(shape):
      raise ValueError(f"{mod}/{name}: expected shape {shape}, got {tuple(t.shape)}")
    chunks.append(t.reshape(-1).to(torch.float32))
  return Params(cfg, torch.cat(chunks).contiguous().to(device))


def flatten(cfg: FlowConfig, params, device) -> torch.Tensor:
  """Flat float32 device tensor for the C ABI; zero-copy for `Params`."""
  if isinstance(params, Params):
    if params.cfg != cfg:
      raise ValueError("params were built for a different FlowConfig")
    flat = params.flat
  elif torch.is_tensor(params):
    flat = params.reshape(-1)
  else:
    flat = from_tree(cfg, params, device).flat
  if flat.device != torch.device(device) or flat.dtype != torch.float32:
    flat = flat.to(device=device, dtype=torch.float32)
  return flat.contiguous()
