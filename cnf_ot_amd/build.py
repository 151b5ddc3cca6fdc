"""Builds cnf_ot_amd/lib/libcnf_ot_amd.so for gfx950 with hipcc (in-tree).

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box
with the repo snapshot.  ``python -m cnf_ot_amd.build`` or
``__graft_entry__.build()``.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcnf_ot_amd.so")
SOURCES = ["cnf_flow.hip", "cnf_grad.hip"]
HEADERS = ["cnf_device.h", "cnf_common.h", "cnf_backward.h", "cnf_pwl.h", "cnf_pwl_build.h",
           os.path.join("..", "..", "include", "cnf_ot_amd.h")]
VARIANT_PATH = os.path.join(LIB_DIR, "BUILD_VARIANT")     # "full" or "minimal": what the .so in tree contains
ARCH = "gfx950"


def _hipcc():
  for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
    if cand and os.path.exists(cand):
      return cand
  raise RuntimeError("hipcc not found (need ROCm's hipcc to build the gfx950 kernels)")


def _variant() -> str:
  try:
    with open(VARIANT_PATH) as f:
      return f.read().strip()
  except OSError:
    return ""


def is_stale(minimal: bool = False) -> bool:
  if not os.path.exists(LIB_PATH):
    return True
  if not minimal and _variant() != "full":     # a quick-iteration build must never pass for the full one
    return True
  t = os.path.getmtime(LIB_PATH)
  deps = [os.path.join(SRC_DIR, f) for f in SOURCES + HEADERS]
  return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, minimal: bool = False, verbose: bool = False) -> str:
  """Compile every HIP source into one shared library.  `minimal` builds only
  the default (hidden_size=16, num_bins=5) kernels -- for quick iteration."""
  if not force and not is_stale(minimal):
    return LIB_PATH
  os.makedirs(LIB_DIR, exist_ok=True)
  cmd = [_hipcc(), "-O3", f"--offload-arch={ARCH}", "-std=c++17", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function"]
  if minimal:
    cmd.append("-DCNF_MINIMAL_CONFIGS")
  cmd += os.environ.get("CNF_EXTRA_FLAGS", "").split()      # experiment switches (scripts/): never set by the product
  cmd += [os.path.join(SRC_DIR, s) for s in SOURCES]
  tmp = LIB_PATH + ".tmp"
  cmd += ["-o", tmp]
  if verbose:
    print(" ".join(cmd), flush=True)
  res = subprocess.run(cmd, capture_output=True, text=True)
  if res.returncode != 0:
    raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
  os.replace(tmp, LIB_PATH)
  with open(VARIANT_PATH, "w") as f:
    f.write("minimal\n" if minimal else "full\n")
  return LIB_PATH


if __name__ == "__main__":
  path = build(force="--force" in sys.argv, minimal="--minimal" in sys.argv, verbose=True)
  print("built", path)
