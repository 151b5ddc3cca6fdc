#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of a HIP source (hipcc -S --cuda-device-only):
  python scripts/kernel_resources.py cnf_ot_amd/csrc/cnf_flow.hip [--full]"""
import os, re, subprocess, sys, tempfile
src = sys.argv[1]
flags = ([] if "--full" in sys.argv else ["-DCNF_MINIMAL_CONFIGS"]) + os.environ.get("CNF_EXTRA_FLAGS", "").split()
out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", *flags, "-S", "--cuda-device-only",
                src, "-o", out], check=True, capture_output=True)
txt = open(out).read()
names = []
for b in txt.split("  - .agpr_count:")[1:]:
  g = lambda k: re.search(r"\." + k + r":\s+(\S+)", b).group(1)
  names.append((g("name"), g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("vgpr_spill_count")))
dem = subprocess.run(["c++filt"], input="\n".join(n[0] for n in names), capture_output=True, text=True).stdout.split("\n")
for (n, v, s, p, sp), d in zip(names, dem):
  print(f"{d[:110]:110s} vgpr {v:>4} sgpr {s:>4} scratch {p:>5} spills {sp:>3}")
print("asm:", out)
