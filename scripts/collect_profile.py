#!/usr/bin/env python3
"""Copies the rocprofv3 summaries of gpurun_out/prof_<tag>/ (scratch) into
profiles/<tag>/ (tracked): the --stats kernel table and a JSON of the PMC
counters of the flow kernels, per launch.  Usage: scripts/collect_profile.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag):
  src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
  dst = os.path.join(ROOT, "profiles", tag)
  os.makedirs(dst, exist_ok=True)
  stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
  if stats:
    shutil.copy(stats[0], os.path.join(dst, "kernel_stats.csv"))
  stats = glob.glob(os.path.join(src, "trace_metric", "*", "*_kernel_stats.csv"))
  if stats:
    shutil.copy(stats[0], os.path.join(dst, "kernel_stats_metric_only.csv"))
  if os.path.exists(os.path.join(src, "bench_line.json")):
    shutil.copy(os.path.join(src, "bench_line.json"), os.path.join(dst, "bench_line.json"))
  out = {}
  for path in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
      for r in csv.DictReader(f):
        if "cnf::" in r["Kernel_Name"] and "fill_normal" not in r["Kernel_Name"] and "prepare" not in r["Kernel_Name"]:
          agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kern, ctrs in agg.items():
      for name, v in ctrs.items():
        out.setdefault(kern, {})[name] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
  with open(os.path.join(dst, "pmc_summary.json"), "w") as f:
    json.dump({"source": f"gpurun_out/prof_{tag} (scripts/gpu_profile.sh {tag})", "kernels": out}, f, indent=1)
  # HBM traffic of the metric kernel, corrected as MI355X_MICROARCH.md prescribes:
  # FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of a
  # coalesced streaming read (checked here against the known algorithmic bytes).
  for kern, ctrs in out.items():
    if "flow_pwl_kernel" in kern and "false" in kern and "FETCH_SIZE" in ctrs and "WRITE_SIZE" in ctrs and len(sys.argv) > 2:
      spl = int(sys.argv[2])
      rd, wr = ctrs["FETCH_SIZE"]["mean"] * 1024 * 2, ctrs["WRITE_SIZE"]["mean"] * 1024
      with open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w") as f:
        json.dump({"kernel": kern, "samples_per_launch": spl, "read_bytes": rd, "write_bytes": wr,
                   "bytes_per_launch": rd + wr, "algorithmic_bytes": 20 * spl,
                   "source": f"profiles/{tag}/pmc_summary.json: FETCH_SIZE*1024*2 + WRITE_SIZE*1024 "
                             "(separate --pmc passes; gfx950 FETCH_SIZE x2 correction)"}, f, indent=1)
      print(f"hbm traffic/launch: read {rd/1e6:.1f} MB + write {wr/1e6:.1f} MB vs algorithmic {20*spl/1e6:.1f} MB")
  for kern, ctrs in out.items():
    print(kern)
    for name, v in sorted(ctrs.items()):
      print(f"   {name:34s} mean={v['mean']:.5g} (n={v['launches']})")


if __name__ == "__main__":
  main(sys.argv[1] if len(sys.argv) > 1 else "r01")
