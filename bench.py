#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X.

metric : MC samples/sec through flow fwd + log|det J| (sample_and_log_prob,
         the reference's conditional.py:353-402) at dim=2, batch=65536.
step   : one pass of the hot path over one batch of 65 536 synthetic base-noise
         samples (config 2 of BASELINE.json: OT free, dim=2) with its own
         condition t; inputs resident in HBM before the timed region; outputs
         y[65536,2] and log_prob[65536] written to HBM for every step.
launch : steps are issued as fused multi-slice launches (`--slices-per-launch`
         steps per launch, each slice with its own c) -- the shape of the
         reference's evaluator cnf_ot/utils.py:311-340 (10 000 slices x 65 536);
         `per_call` in the JSON line is the same work issued one launch per step.
N GPUs : one process per GPU (torch.distributed, backend nccl = RCCL); every
         step's batch is sample-sharded in contiguous blocks of 65536/N
         (SURVEY.md 8e); no data-path collective; total work fixed => "strong".

  python bench.py --gpus 1 --steps 16384 --warmup 1024
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

METRIC = "MC samples/sec through flow fwd+log|detJ| at dim=2, batch=65536"
DIM = 2
BATCH = 65536
# algorithmic work per sample per flow pass, fp32, default net
FLOP_PER_SAMPLE = 2176 + 400          # conditioner MLP + splines (the reference's formulation)
# what the dim-2 table path executes: the conditioner is read from exact piecewise-linear tables,
# 16 FMAs per layer instead of the 544-FMA MLP
EXECUTED_FLOP_PER_SAMPLE = 2 * 32 + 400
BYTES_PER_SAMPLE = 4 * (2 * DIM + 1)  # x in, y out, log_prob out (c is per slice)
PEAK_FP32_TFLOPS = 157.3              # MI355X_MICROARCH.md: fp32 vector == fp32 MFMA peak
PEAK_HBM_GBS = 8000.0


def parse_args():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=65536)
  ap.add_argument("--warmup", type=int, default=4096)
  ap.add_argument("--slices-per-launch", type=int, default=256)
  ap.add_argument("--param-scale", type=float, default=0.2)
  ap.add_argument("--shard", choices=("slices", "samples"), default="slices",
                  help="multi-GPU partition of the samples x time-slices grid (see main())")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--cpu-seconds", type=float, default=12.0)
  ap.add_argument("--no-per-call", action="store_true")
  ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); 'gloo' + "
                  "--share-device rehearses the multi-rank path on a one-GPU box")
  ap.add_argument("--share-device", action="store_true", help="all ranks use cuda:0 (rehearsal only)")
  ap.add_argument("--cpu-threads", type=int, default=0, help="OpenMP threads of the CPU baseline "
                  "(0: min(16, affinity): a one-GPU box's CPU share)")
  return ap.parse_args()


def cpu_baseline(params64, seconds, threads=0):
  """The oracle (float64 C restatement, OpenMP over the host cores) timed on a
  bounded sample of the same workload.  A reported baseline, not the target."""
  import oracle
  ocfg = oracle.OracleConfig(D=DIM)
  oracle.build_library()
  if threads <= 0:
    threads = min(16, len(os.sched_getaffinity(0)))
  oracle.set_num_threads(threads)
  rng = np.random.default_rng(0)
  noise = rng.normal(size=(BATCH, DIM))
  oracle.sample_logprob(ocfg, params64, noise, [0.5])          # warm-up
  t0 = time.perf_counter()
  n = 0
  while True:
    oracle.sample_logprob(ocfg, params64, noise, [n / 64.0 % 1.0])
    n += 1
    dt = time.perf_counter() - t0
    if dt >= seconds or n >= 4096:
      break
  return {
    "value": n * BATCH / dt, "unit": "samples/s", "cores": oracle.num_threads(),
    "kind": "port",
    "sample": f"{n} batches of {BATCH} (float64 C oracle, OpenMP, {dt:.1f} s)",
  }


def main():
  args = parse_args()
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if world != args.gpus:
    if world == 1 and args.gpus > 1:
      raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    args.gpus = world
  assert torch.cuda.is_available(), "bench.py needs an MI355X (ROCm) device"
  dev = torch.device("cuda", 0 if args.share_device else local_rank)
  torch.cuda.set_device(dev)
  dist = None
  if world > 1:
    import torch.distributed as dist
    if args.backend == "nccl":
      dist.init_process_group("nccl", device_id=dev)
    else:
      dist.init_process_group(args.backend)

  from cnf_ot_amd import FlowConfig, FlowEngine, Params

  # A launch fuses many steps (batches of 65 536 samples, each with its own condition t).  The
  # samples x time-slices grid of a launch is sharded over the ranks by whole slices: a global launch
  # covers S * world steps and every rank processes S of them (the per-rank launch has the same shape
  # at every N; nothing is exchanged -- sampling has no reduction).  --shard samples splits every
  # batch instead (what the loss functions of cnf_ot_amd.applications do, where slices can be few).
  if args.shard == "samples":
    if BATCH % world != 0 or (BATCH // world) % 256 != 0:
      raise SystemExit(f"batch {BATCH} does not shard into 256-sample tiles over {world} GPUs")
    b_local, n_share, my = BATCH // world, 1, 0
  else:
    b_local, n_share, my = BATCH, world, rank
  S = max(1, min(args.slices_per_launch, -(-args.steps // n_share)))

  cfg = FlowConfig(dim=DIM)
  params = Params.random(cfg, args.param_scale, seed=42, device=dev)
  eng = FlowEngine(cfg, dev).load(params)

  # synthetic inputs, resident in HBM: S distinct slices per rank; global slice g = my * S + s of a
  # launch is samples [g*BATCH, +BATCH) of the seed-42 Philox stream (this rank's part of it)
  noise = torch.empty(S, b_local, DIM, device=dev)
  first = 0 if args.shard == "slices" else rank * b_local
  for s in range(S):
    noise[s] = eng.normal(42, b_local, first_sample=(my * S + s) * BATCH + first)
  t_slices = torch.linspace(0.0, 1.0, S * n_share, device=dev)[my * S:(my + 1) * S].contiguous()
  y = torch.empty(S * b_local, DIM, device=dev)
  lp = torch.empty(S * b_local, device=dev)
  noise_flat = noise.view(S * b_local, DIM)

  def run(n_steps, events=None):
    # n_steps global steps; a global launch takes up to S * n_share of them, this rank its share
    done = 0
    while done < n_steps:
      g = min(S * n_share, n_steps - done)
      s = g // n_share + (1 if my < g % n_share else 0)
      done += g
      if s == 0:
        continue
      if events is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
      eng.sample_logprob(noise_flat[:s * b_local], t_slices[:s], out=y[:s * b_local], logp_out=lp[:s * b_local])
      if events is not None:
        e1.record()
        events.append((e0, e1, s))

  def barrier():
    torch.cuda.synchronize()
    if dist is not None:
      dist.barrier()
    torch.cuda.synchronize()

  run(max(args.warmup, 1) if args.warmup > 0 else 0)
  barrier()
  events = []
  t0 = time.perf_counter()
  run(args.steps, events)
  torch.cuda.synchronize()
  elapsed = time.perf_counter() - t0
  if dist is not None:
    dist.barrier()
    t = torch.tensor([elapsed], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
  assert torch.isfinite(lp).all(), "non-finite log_prob in the bench output"

  # dominant kernel: average launch duration from HIP events on the launch stream
  full = [(a.elapsed_time(b) * 1e-3, s) for a, b, s in events if s == S] or \
         [(a.elapsed_time(b) * 1e-3, s) for a, b, s in events] or [(float("nan"), 0)]    # a rank with no step
  k_dur = sum(d for d, _ in full) / len(full)
  k_samples = full[0][1] * b_local
  achieved_tflops = FLOP_PER_SAMPLE * k_samples / k_dur / 1e12
  executed_tflops = EXECUTED_FLOP_PER_SAMPLE * k_samples / k_dur / 1e12
  achieved_gbs = BYTES_PER_SAMPLE * k_samples / k_dur / 1e9

  value = args.steps * BATCH / elapsed
  line = {
    "metric": METRIC, "value": value, "unit": "samples/s", "n_gpus": world,
    "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
    "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
    "dtype": "f32", "data": "synthetic",
    "config": {
      "workload": "configs[1]: OT free dim=2, batch=65536, sample_and_log_prob (RQS fwd + log|detJ|), "
                  "L=2 H=16 M=2 K=5, params N(0,%.2f^2) seed 42" % args.param_scale,
      "batch": BATCH, "dim": DIM, "batch_per_gpu": b_local,
      "slices_per_launch": S, "launches": len(events),
      "parallelism": (f"slice-shard x{world} (each rank {S} whole batches per launch)" if args.shard == "slices"
                      else f"sample-shard x{world}"),
    },
    "roofline": {
      # HBM: SURVEY.md 8(d)'s algorithmic bytes (20 B per sample: x in, y out, log_prob out) -- the one bound a
      # reformulation cannot move.  The ALU views are beside it.
      "bound": "hbm", "achieved": achieved_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
      "frac": achieved_gbs / PEAK_HBM_GBS, "traffic": None,
      "kernel": "cnf::pwl_build_kernel + cnf::flow_pwl_kernel<5,false,true> (one cnf_sample_logprob call: "
                "piecewise-linear conditioner tables + packed fp32 VALU flow, 2 samples/lane)",
      "launch_ms": k_dur * 1e3,
      "samples_per_launch": k_samples, "bytes_per_sample": BYTES_PER_SAMPLE,
      "alu_executed": {"flop_per_sample": EXECUTED_FLOP_PER_SAMPLE, "achieved": executed_tflops,
                       "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": executed_tflops / PEAK_FP32_TFLOPS},
      "alu_reference_formulation": {"flop_per_sample": FLOP_PER_SAMPLE, "achieved": achieved_tflops,
                                    "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                                    "frac": achieved_tflops / PEAK_FP32_TFLOPS},
      "note": "SURVEY.md 8(d) derived an fp32-ALU bound of 60 G samples/s from the reference formulation (2576 flop "
              "per sample: 2-16-16-16 MLP conditioner + splines).  The kernel reads the same conditioner from exact "
              "piecewise-linear tables and executes ~464 flop per sample, so that figure is no longer a bound "
              "(alu_reference_formulation.frac can exceed 1); the bound left is HBM.  What limits the kernel is "
              "VALU issue: 797 instructions per 128-sample wave-tile (DESIGN.md 5.1d).",
    },
  }
  pmc = os.path.join(ROOT, "profiles", "hbm_traffic.json")
  if os.path.exists(pmc):
    with open(pmc) as f:
      rec = json.load(f)
    if rec.get("samples_per_launch"):
      line["roofline"]["traffic"] = rec["bytes_per_launch"] * k_samples / rec["samples_per_launch"]
      line["roofline"]["traffic_source"] = rec.get("source", "profiles/hbm_traffic.json")

  if not args.no_per_call and rank == 0 and world == 1:
    n_calls = 200
    for _ in range(20):
      eng.sample_logprob(noise_flat[:b_local], t_slices[:1], out=y[:b_local], logp_out=lp[:b_local])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_calls):
      eng.sample_logprob(noise_flat[:b_local], t_slices[i % S:i % S + 1], out=y[:b_local], logp_out=lp[:b_local])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    line["per_call"] = {"ms_per_step": dt / n_calls * 1e3, "value": n_calls * BATCH / dt,
                        "note": "one launch per 65536-sample step, eager (MLP kernel: a single batch "
                                "does not amortise building the tables)"}
    # the same launch-bound loop captured once into a HIP graph and replayed (entry points only enqueue)
    try:
      side = torch.cuda.Stream(device=dev)
      side.wait_stream(torch.cuda.current_stream(dev))
      graph = torch.cuda.CUDAGraph()
      with torch.cuda.graph(graph, stream=side):
        for i in range(n_calls):
          eng.sample_logprob(noise_flat[:b_local], t_slices[i % S:i % S + 1], out=y[:b_local], logp_out=lp[:b_local])
      graph.replay(); torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(5):
        graph.replay()
      torch.cuda.synchronize()
      dt = (time.perf_counter() - t0) / 5
      line["per_call"]["hip_graph"] = {"ms_per_step": dt / n_calls * 1e3, "value": n_calls * BATCH / dt,
                                       "note": f"{n_calls} calls captured in one graph, replayed"}
    except Exception as exc:      # reported, never fatal for the metric
      line["per_call"]["hip_graph"] = {"error": repr(exc)[:200]}

  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    line["cpu_baseline"] = cpu_baseline(params.flat.cpu().double().numpy(), args.cpu_seconds, args.cpu_threads)
  elif rank == 0:
    line["cpu_baseline"] = None

  if rank == 0:
    print(json.dumps(line), flush=True)
  if dist is not None:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
