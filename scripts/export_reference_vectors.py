#!/usr/bin/env python3
"""Run ON A BOX THAT HAS JAX + the reference checkout (pip install jax distrax dm-haiku; cd jiaxi98/cnf_ot).
It cannot run in the build container (no jax): this is the one missing pin of the oracle.
Writes reference_d{D}.npz: parameters BY HAIKU NAME ("module/name"), inputs, and the reference's own outputs in
float64; drop the files into tests/golden/ and tests/test_oracle_flow.py::test_reference_vectors_if_present
compares the oracle (and, on a GPU box, the HIP path) with them."""
import numpy as np, jax, jax.numpy as jnp, haiku as hk
from cnf_ot.models.flows import RQSFlow
jax.config.update("jax_enable_x64", True)
for D in (2, 10):
  model = hk.without_apply_rng(hk.multi_transform(
    RQSFlow(event_shape=(D,), num_layers=2, hidden_sizes=[16, 16], num_bins=5, periodized=False)))
  params = model.init(jax.random.PRNGKey(0), jnp.zeros((1, D)), jnp.zeros((1,)))
  rng = np.random.default_rng(42)     # non-trivial parameters: N(0, 0.2^2) on every leaf, float32-representable
  params = {m: {n: jnp.asarray(rng.normal(0, 0.2 if D == 2 else 0.12, a.shape).astype(np.float32), a.dtype)
                for n, a in sorted(d.items())} for m, d in sorted(params.items())}
  x = jnp.asarray(rng.normal(size=(4096, D)).astype(np.float32), jnp.float64)
  c = jnp.asarray([0.37])
  y = model.apply.forward(params, x, c)                     # base -> data
  out = {f"{m}/{n}": np.asarray(a) for m, d in params.items() for n, a in d.items()}
  key = jax.random.PRNGKey(7)          # the base draw of conditional.py:378,399, to pin cnf_fill_normal_threefry
  out.update(key_words=np.asarray(jax.random.key_data(key)), noise=np.asarray(jax.random.normal(key, (1000, D))))
  out.update(x=np.asarray(x), c=np.asarray(c), y=np.asarray(y),
             log_prob_y=np.asarray(model.apply.log_prob(params, y, c)),
             x_back=np.asarray(model.apply.inverse(params, y, c)))
  np.savez(f"reference_d{D}.npz", **out)
  print("wrote", f"reference_d{D}.npz")
