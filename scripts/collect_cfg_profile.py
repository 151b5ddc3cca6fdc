#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (scripts/gpu_pmc_cfg.sh) -> profiles/<tag>/{kernel_stats.csv, pmc_summary.json}: per kernel
the mean counter values per launch and, derived from them, what north_star asks for -- MFMA busy share and achieved
fp32-MFMA TFLOP/s against the 157.3 TFLOP/s peak (SQ_INSTS_VALU_MFMA_MOPS_F32 counts 512-flop units on gfx950:
MI355X_MICROARCH.md; the derivation is printed so it can be checked against the kernel's known MFMA count)."""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK_TF, CLK_GHZ, SIMDS = 157.3, 2.4, 1024


def main(tag):
  src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
  dst = os.path.join(ROOT, "profiles", tag)
  os.makedirs(dst, exist_ok=True)
  dur = {}
  stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
  if stats:
    shutil.copy(stats[0], os.path.join(dst, "kernel_stats.csv"))
    with open(stats[0]) as f:
      for r in csv.DictReader(f):
        dur[r["Name"]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])}
  log = os.path.join(src, "trace.log")
  if os.path.exists(log):
    with open(log) as f:
      lines = [l for l in f if " ms per call" in l]
    with open(os.path.join(dst, "step_times.txt"), "w") as f:
      f.writelines(lines)
  out = {}
  for path in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
      for r in csv.DictReader(f):
        if "cnf::" in r["Kernel_Name"]:
          agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kern, ctrs in agg.items():
      for name, v in ctrs.items():
        out.setdefault(kern, {})[name] = {"launches": len(v), "mean": sum(v) / len(v)}
  derived = {}
  for kern, c in out.items():
    g = lambda n: c.get(n, {}).get("mean")
    d = {}
    if kern in dur:
      d["avg_launch_us"] = dur[kern]["avg_ns"] / 1e3
      d["share_of_gpu_time_pct"] = dur[kern]["pct"]
    busy, mfma_busy = g("SQ_BUSY_CYCLES"), g("SQ_VALU_MFMA_BUSY_CYCLES")
    wave_cyc, valu_act = g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_VALU")
    gui = g("GRBM_GUI_ACTIVE")
    mops = g("SQ_INSTS_VALU_MFMA_MOPS_F32")
    if gui and mfma_busy is not None:
      # SQ_VALU_MFMA_BUSY_CYCLES: cycles (32 per v_mfma_f32_16x16x4_f32) summed over every SIMD of the chip;
      # GRBM_GUI_ACTIVE: the kernel's cycles summed over the 8 XCDs (MI355X_MICROARCH.md)
      d["mfma_busy_pct_of_simd_cycles"] = 100.0 * mfma_busy / (gui / 8.0 * SIMDS)
    if mops is not None and kern in dur and dur[kern]["avg_ns"] > 0:
      d["mfma_flop_per_launch (MOPS x 512)"] = mops * 512
      d["mfma_tflops"] = mops * 512 / dur[kern]["avg_ns"] / 1e3
      d["mfma_frac_of_fp32_peak"] = d["mfma_tflops"] / PEAK_TF
    if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
      d["valu_insts_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES")
    if g("SQ_INSTS_MFMA") and g("SQ_WAVES"):
      d["mfma_insts_per_wave"] = g("SQ_INSTS_MFMA") / g("SQ_WAVES")
    if g("SQ_INSTS_LDS") and g("SQ_WAVES"):
      d["lds_insts_per_wave"] = g("SQ_INSTS_LDS") / g("SQ_WAVES")
    if wave_cyc and valu_act:
      d["valu_active_share_of_wave_cycles"] = valu_act / wave_cyc
    if wave_cyc and g("SQ_WAIT_ANY"):
      d["wait_any_share_of_wave_cycles"] = g("SQ_WAIT_ANY") / wave_cyc
    if wave_cyc and g("SQ_WAIT_INST_LDS"):
      d["wait_lds_share_of_wave_cycles"] = g("SQ_WAIT_INST_LDS") / wave_cyc
    derived[kern] = d
  with open(os.path.join(dst, "pmc_summary.json"), "w") as f:
    json.dump({"source": f"gpurun_out/prof_{tag} (scripts/gpu_pmc_cfg.sh)", "peak_fp32_tflops": PEAK_TF,
               "derived": derived, "kernels": out}, f, indent=1)
  for kern in sorted(derived, key=lambda k: -derived[k].get("share_of_gpu_time_pct", 0)):
    print(kern[:150])
    for k, v in derived[kern].items():
      print(f"   {k:48s} {v:.5g}")
    for name in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD",
                 "SQ_INSTS_VMEM_WR", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
      if name in out[kern]:
        print(f"   {name:48s} {out[kern][name]['mean']:.5g}")


if __name__ == "__main__":
  main(sys.argv[1])
