"""ctypes binding of libcnf_ot_amd.so -- the C ABI declared in
include/cnf_ot_amd.h.  This is the stub a cnf_ot maintainer would add to call
the MI355X engine from Python.

There is NO CPU fallback: if the HIP library is missing or does not load,
importing the product path fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcnf_ot_amd.so")

CNF_OK = 0
CNF_ERR_INVALID = -22
CNF_ERR_UNSUPPORTED = -95
CNF_ERR_NOMEM = -12
CNF_ERR_HIP = -5


class CnfConfig(ctypes.Structure):
  _fields_ = [
    ("dim", ctypes.c_int32), ("num_layers", ctypes.c_int32),
    ("hidden_size", ctypes.c_int32), ("mlp_num_layers", ctypes.c_int32),
    ("num_bins", ctypes.c_int32),
    ("range_min", ctypes.c_float), ("range_max", ctypes.c_float),
    ("min_bin_size", ctypes.c_float), ("min_knot_slope", ctypes.c_float),
    ("periodized", ctypes.c_int32),
  ]


class CnfLossSpec(ctypes.Structure):
  _fields_ = [
    ("kind", ctypes.c_int32), ("subtype", ctypes.c_int32),
    ("dt", ctypes.c_float), ("dx", ctypes.c_float), ("coef", ctypes.c_float),
    ("a", ctypes.c_float), ("T", ctypes.c_float), ("beta", ctypes.c_float),
  ]


TERM_KINETIC, TERM_KINETIC_SCORE, TERM_FLOW_MATCHING, TERM_POTENTIAL, TERM_REVERSE_KL, TERM_NEG_LOGPROB = range(6)
POTENTIALS = {"quadratic": 0, "double_well": 1, "obstacle": 2}
DRIFTS = {"ou": 0, "gradient": 1, "smile": 1, "nongradient": 2, "lorenz": 3}

# name -> (restype, argtypes); every symbol include/cnf_ot_amd.h declares
_P = ctypes.c_void_p
_I64 = ctypes.c_int64
_U64 = ctypes.c_uint64
_CFG = ctypes.POINTER(CnfConfig)
SYMBOLS = {
  "cnf_config_default": (None, [_CFG, ctypes.c_int32]),
  "cnf_param_count": (_I64, [_CFG]),
  "cnf_model_create": (ctypes.c_int, [_CFG, ctypes.POINTER(_P)]),
  "cnf_model_destroy": (None, [_P]),
  "cnf_model_set_params": (ctypes.c_int, [_P, _P, _P]),
  "cnf_model_reserve": (ctypes.c_int, [_P, _P, _I64]),
  "cnf_model_reserved": (_I64, [_P, _P]),
  "cnf_model_table_bytes": (_I64, [_P]),
  "cnf_model_last_path": (ctypes.c_int, [_P]),
  "cnf_model_set_precise": (ctypes.c_int, [_P, ctypes.c_int]),
  "cnf_forward_logdet": (ctypes.c_int, [_P, _P, _P, _I64, _P, _P, _I64, _P]),
  "cnf_inverse_logdet": (ctypes.c_int, [_P, _P, _P, _I64, _P, _P, _I64, _P]),
  "cnf_log_prob": (ctypes.c_int, [_P, _P, _P, _I64, _P, _I64, _P]),
  "cnf_sample_logprob": (ctypes.c_int, [_P, _P, _P, _I64, _P, _P, _I64, _P]),
  "cnf_sample_logprob_seeded": (ctypes.c_int, [_P, _U64, _I64, _I64, _P, _I64, _P, _P, _I64, _P]),
  "cnf_forward_logdet_f64": (ctypes.c_int, [_P, _P, _P, _I64, _P, _P, _I64, _P]),
  "cnf_inverse_logdet_f64": (ctypes.c_int, [_P, _P, _P, _I64, _P, _P, _I64, _P]),
  "cnf_log_prob_f64": (ctypes.c_int, [_P, _P, _P, _I64, _P, _I64, _P]),
  "cnf_sample_logprob_f64": (ctypes.c_int, [_P, _P, _P, _I64, _P, _P, _I64, _P]),
  "cnf_fill_normal": (ctypes.c_int, [_U64, _U64, _I64, _P, _P]),
  "cnf_fill_normal_threefry": (ctypes.c_int, [ctypes.c_uint32, ctypes.c_uint32, _U64, _U64, _I64, _P, _P, _P]),
  "cnf_loss_terms": (ctypes.c_int, [_P, ctypes.POINTER(CnfLossSpec), _P, ctypes.c_int, _P, _I64, _I64, _P, _P]),
  "cnf_loss_terms_seeded": (ctypes.c_int, [_P, ctypes.POINTER(CnfLossSpec), _U64, _I64, _I64, _P, _I64, _I64, _P, _P]),
  "cnf_grad_supported": (ctypes.c_int, [_CFG]),
  "cnf_grad_enable": (ctypes.c_int, [_P, _I64]),
  "cnf_loss_terms_grad": (ctypes.c_int, [_P, ctypes.POINTER(CnfLossSpec), _P, ctypes.c_int, _P, _I64, _I64,
                                         ctypes.c_float, _P, _P, _P, _P]),
  "cnf_loss_terms_grad_multi": (ctypes.c_int, [_P, ctypes.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
  "cnf_input_vjp": (ctypes.c_int, [_P, ctypes.c_int, _P, _P, _I64, _P, _P, _P, _I64, _P]),
  "cnf_pass_vjp": (ctypes.c_int, [_P, ctypes.c_int, _P, _P, _I64, _P, _P, _P, _P, _P, _I64, _P]),
  "cnf_neg_logprob_vjp": (ctypes.c_int, [_P, _P, _P, _I64, ctypes.c_float, _P, _P, _P, _I64, _P]),
  "cnf_kinetic_potential_vjp": (ctypes.c_int, [_P, _P, _I64, _P, ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_int32,
                                               ctypes.c_float, ctypes.c_float, _P, _P, _P, _P, _P, _P]),
  "cnf_logprob_fd": (ctypes.c_int, [_P, _P, _P, _I64, ctypes.c_float, _P, _I64, _P]),
  "cnf_logprob_fd_vjp": (ctypes.c_int, [_P, _P, _P, _I64, ctypes.c_float, _P, _P, _P, _P, _I64, _P]),
  "cnf_score_fd_vjp": (ctypes.c_int, [_P, _P, _P, _I64, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int32,
                                      ctypes.c_float, ctypes.c_float, _P, _P, _P, _P, _I64, _P]),
  "cnf_score_residual": (ctypes.c_int, [_P, _P, _I64, _I64, ctypes.c_int32, ctypes.c_float, ctypes.c_float,
                                        ctypes.c_int32, ctypes.c_float, ctypes.c_float, _P, _P, _P, _P]),
  "cnf_term_residual": (ctypes.c_int, [ctypes.c_int32, _P, _P, _I64, _I64, ctypes.c_int32, ctypes.c_int32, ctypes.c_float,
                                       ctypes.c_float, _P, _P, _P, _P]),
  "cnf_rkl_residual": (ctypes.c_int, [_P, _P, _I64, ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                      ctypes.c_float, _P, _P, _P, _P]),
  "cnf_adam_step": (ctypes.c_int, [_P, _P, _P, _P, _I64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                   ctypes.c_float, _I64, _P]),
  "cnf_step_begin": (ctypes.c_int, [_P, _P]),
  "cnf_fill_normal_dev": (ctypes.c_int, [_P, _U64, _I64, _P, _P]),
  "cnf_fill_uniform_dev": (ctypes.c_int, [_P, _U64, _I64, ctypes.c_float, _P, _P]),
  "cnf_mixture_source_dev": (ctypes.c_int, [_P, _U64, _I64, _P, _P, _P, _P]),
  "cnf_adam_step_dev": (ctypes.c_int, [_P, _P, _P, _P, _I64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                       ctypes.c_float, _P, _P]),
  "cnf_weighted_sum": (ctypes.c_int, [_P, _P, _I64, _P, _P]),
  "cnf_strerror": (ctypes.c_char_p, [ctypes.c_int]),
  "cnf_build_arch": (ctypes.c_char_p, []),
  "cnf_config_supported": (ctypes.c_int, [_CFG]),
}
# test / measurement knobs: include/cnf_ot_amd_debug.h (not part of the drop-in boundary)
_INTERNAL = {
  "cnf_model_set_fast_math": (ctypes.c_int, [_P, ctypes.c_int]),
  "cnf_model_set_samples_per_lane": (ctypes.c_int, [_P, ctypes.c_int]),
  "cnf_model_set_mfma": (ctypes.c_int, [_P, ctypes.c_int]),
  "cnf_model_set_pwl": (ctypes.c_int, [_P, ctypes.c_int]),
  "cnf_model_set_dpar": (ctypes.c_int, [_P, ctypes.c_int]),
  "cnf_model_set_profiling": (ctypes.c_int, [_P, ctypes.c_int]),
  "cnf_model_read_profile": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                            ctypes.POINTER(_I64), ctypes.POINTER(_I64)]),
}
PATH_NAMES = {0: "none", 1: "mlp1", 2: "mlp2", 3: "mfma", 4: "tables", 5: "loss_mlp", 6: "loss_tables", 7: "f64", 8: "detect", 9: "dpar"}

_lib = None


class CnfError(RuntimeError):
  def __init__(self, code, where):
    self.code = code
    msg = lib().cnf_strerror(code).decode() if _lib is not None else str(code)
    super().__init__(f"{where}: {msg} ({code})")


def lib():
  """Load the HIP library (once).  Raises if it is absent: no fallback."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise ImportError(
      f"{LIB_PATH} is missing: the cnf_ot_amd product path is hand-written HIP "
      "for gfx950 and has no CPU fallback. Build it with "
      "`python -m cnf_ot_amd.build` (needs hipcc).")
  handle = ctypes.CDLL(LIB_PATH)
  for table in (SYMBOLS, _INTERNAL):
    for name, (res, args) in table.items():
      fn = getattr(handle, name)   # AttributeError if the export is missing
      fn.restype = res
      fn.argtypes = args
  _lib = handle
  return _lib


def check(code, where):
  if code != CNF_OK:
    raise CnfError(code, where)
