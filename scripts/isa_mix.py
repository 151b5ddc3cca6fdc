#!/usr/bin/env python3
"""Static instruction mix of one kernel of the assembly scripts/kernel_resources.py leaves in /tmp:
  python scripts/isa_mix.py /tmp/cnf_grad.hip.s _ZN3cnf10vjp_kernelILb1ELb1ELi0EE [--blocks]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2]) and ":" in l and not l.startswith(" ")][0]
end = [i for i in range(start, len(lines)) if "s_endpgm" in lines[i]][0]
blocks, cur = [], ["entry", []]
for l in lines[start + 1:end]:
  t = l.strip()
  m = re.match(r"^(\.LBB\d+_\d+):", t)
  if m:
    blocks.append(cur); cur = [m.group(1), []]; continue
  if not t or t.startswith(";") or t.startswith("."):
    continue
  cur[1].append(t.split()[0])
blocks.append(cur)
def row(name, ins):
  c = collections.Counter(ins)
  g = lambda p: sum(v for k, v in c.items() if k.startswith(p))
  return (f"{name:12s} n={len(ins):5d} mfma={g('v_mfma'):3d} valu={g('v_') - g('v_mfma'):4d} rdl={c['v_readlane_b32']:3d} wrl={c['v_writelane_b32']:3d} "
          f"mov={c['v_mov_b32_e32']:3d} cnd={g('v_cndmask'):3d} swap={g('v_permlane'):3d} ds={g('ds_'):3d} glob={g('global_'):3d} "
          f"salu={g('s_') - c['s_nop'] - c['s_waitcnt']:4d} nop={c['s_nop']:3d} wait={c['s_waitcnt']:3d}")
allins = [i for _, ins in blocks for i in ins]
print(row("TOTAL", allins))
if "--blocks" in sys.argv:
  for name, ins in blocks:
    if len(ins) >= 40:
      print(row(name, ins))
else:
  for k, v in collections.Counter(allins).most_common(30):
    print(f"  {k:28s}{v}")
