"""CPU oracle for the conditional RQS flow path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product (``cnf_ot_amd``) never does.

Parity status: **parity unpinned** in absolute value against distrax (the
reference's spline arithmetic lives in the absent third-party package distrax,
and the reference holds no golden vectors); see ``cnf_oracle_impl.h``.
"""
from .capi import threefry2x32, normal_threefry  # noqa: F401
from .capi import (  # noqa: F401
  OracleConfig,
  build_library,
  load_library,
  param_count,
  forward_logdet,
  inverse_logdet,
  log_prob,
  sample_logprob,
  rqs,
  knots,
  normal,
  num_threads,
  set_num_threads,
)
