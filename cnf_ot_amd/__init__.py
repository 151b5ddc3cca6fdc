"""cnf_ot_amd -- MI355X-native conditional RQS flow engine, a drop-in for the
flow-model call surface of jiaxi98/cnf_ot (cnf_ot/models, consumed by
cnf_ot/mfc/solvers.py).  Hand-written HIP kernels for gfx950 behind a C ABI
(include/cnf_ot_amd.h); no CPU fallback."""
from .params import FlowConfig, Params, param_spec, from_tree  # noqa: F401
from .flows import RQSFlow, FlowModel, FlowEngine, Flow, DeviceRng  # noqa: F401

__all__ = ["RQSFlow", "FlowModel", "FlowEngine", "Flow", "FlowConfig", "Params",
           "param_spec", "from_tree"]
