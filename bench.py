#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X.

metric : MC samples/sec through flow fwd + log|det J| (sample_and_log_prob,
         the reference's conditional.py:353-402) at dim=2, batch=65536.
step   : ONE evaluator-shaped pass of the hot path: `--slices` (default 10 000)
         time-slices x 65 536 base-noise samples, every slice with its own
         condition t, through sample_and_log_prob -- the shape of the
         reference's evaluator cnf_ot/utils.py:311-340 (10 000 slices x 65 536,
         which the reference issues as 20 000 jitted calls; here ONE
         cnf_sample_logprob call per rank).  Inputs are resident in HBM before
         the timed region; y[.,2] and log_prob[.] are written for every sample.
         Warm-up steps are identical calls (same shape, same kernels, table
         workspace reserved before the first of them).
N GPUs : one process per GPU (torch.distributed, backend nccl = RCCL); the
         slices of every step are sharded over the ranks in contiguous blocks
         (samples x slices are independent; sampling has no
         reduction, so there is no data-path collective); total work per step
         is fixed => "strong".
extras : (rank 0, N=1) per-kernel HIP-event times of the same calls (roofline),
         the single-batch call pattern (`per_call`), the dim-10 kernel against
         its ALU roofline, the loss / value_and_grad steps of BASELINE configs
         3-5, and the CPU baseline (the oracle, timed on a bounded sample).
--workload cfg3|cfg4|cfg5 : times `update()` (value_and_grad + Adam, with the
         ONE all-reduce of partial sums + gradient when sharded) instead of the
         headline metric; own metric string.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python bench.py --gpus N --steps K --warmup W          (starts its own N ranks as child processes)
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
FLOP_PER_SAMPLE_D10 = 21888 + 2400
# what the dim-2 table path executes: the conditioner is read from exact piecewise-linear tables,
# 12 FMAs per layer (10 softmax logits + the selected bin's 2 slope logits) instead of the 544-FMA MLP
EXECUTED_FLOP_PER_SAMPLE = 2 * 24 + 400
BYTES_PER_SAMPLE = 4 * (2 * DIM + 1)  # x in, y out, log_prob out (c is per slice)
PEAK_FP32_TFLOPS = 157.3              # MI355X_MICROARCH.md: fp32 vector == fp32 MFMA peak
PEAK_HBM_GBS = 8000.0
KERNEL_NAMES = {
  "tables": "cnf::flow_pwl_kernel<5,false,true,false,289,2> (conditioner from piecewise-linear tables built by "
            "cnf::pwl_build_kernel; packed fp32 VALU flow, 2 samples/lane)",
  "mlp2": "cnf::flow_kernel<16,5,false,true,v2f> (conditioner MLP in packed fp32 VALU, 2 samples/lane)",
  "mlp1": "cnf::flow_kernel<16,5,false,true,float> (conditioner MLP, 1 sample/lane)",
  "mfma": "cnf::flow_kernel<16,5,false,true,.,MFMA> (conditioner on v_mfma_f32_16x16x4_f32)",
}


def parse_args():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=20)
  ap.add_argument("--warmup", type=int, default=5)
  ap.add_argument("--slices", type=int, default=10000,
                  help="time-slices of 65 536 samples per step (the evaluator of utils.py:311-340 has 10 000)")
  ap.add_argument("--param-scale", type=float, default=0.2)
  ap.add_argument("--workload", choices=("sample", "cfg3", "cfg4", "cfg5"), default="sample")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--cpu-seconds", type=float, default=8.0)
  ap.add_argument("--no-extras", action="store_true", help="skip per_call / dim-10 / config 3-5 sections")
  ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); 'gloo' + "
                  "--share-device rehearses the multi-rank path on a one-GPU box")
  ap.add_argument("--share-device", action="store_true", help="all ranks use cuda:0 (rehearsal only)")
  ap.add_argument("--force-dist", action="store_true", help="initialise the process group even for ONE rank (checks "
                  "the RCCL init / all-reduce / barrier path on a one-GPU box)")
  ap.add_argument("--cpu-threads", type=int, default=0, help="OpenMP threads of the CPU baseline (0: the fastest of "
                  "{affinity, cgroup quota, 64, 32, 16}, found by a short calibration)")
  return ap.parse_args()


def _cgroup_cpus():
  try:
    with open("/sys/fs/cgroup/cpu.max") as f:
      q, p = f.read().split()
      if q != "max":
        return max(1, int(int(q) / int(p)))
  except (OSError, ValueError):
    pass
  try:
    with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
      q = int(f.read())
    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
      p = int(f.read())
    if q > 0:
      return max(1, q // p)
  except (OSError, ValueError):
    pass
  return None


def cpu_baseline(params64, seconds, threads=0):
  """The oracle (float64 C restatement, OpenMP over the host cores) timed on a
  bounded sample of the same workload.  A reported baseline, not the target.
  Thread count: BASELINE.md asks for all host cores of the box; a one-GPU box
  exposes the whole host in its affinity mask but schedules a share of it, so
  the candidates {affinity, cgroup quota, 64, 32, 16} are timed briefly and the
  fastest is used (and reported as `cores`)."""
  import oracle
  ocfg = oracle.OracleConfig(D=DIM)
  oracle.build_library()
  rng = np.random.default_rng(0)
  noise = rng.normal(size=(BATCH, DIM))
  affinity = len(os.sched_getaffinity(0))
  tried = {}
  if threads <= 0:
    # ~1 s per candidate: a box that schedules a share of the host lets short bursts run on more cores than it
    # sustains, so a few batches are not enough to rank the candidates
    cands = sorted({c for c in (affinity if affinity <= 128 else None, _cgroup_cpus(), 64, 32, 16)
                    if c and c <= affinity}, reverse=True)
    if affinity > 128:
      print(f"cpu_baseline: affinity_cores not tried: {affinity} > 128 (the box schedules a share of the host; "
            f"candidates {cands})", file=sys.stderr, flush=True)
    for c in cands:
      oracle.set_num_threads(c)
      oracle.sample_logprob(ocfg, params64, noise, [0.5])
      t0, k = time.perf_counter(), 0
      while time.perf_counter() - t0 < 1.0:
        oracle.sample_logprob(ocfg, params64, noise, [0.01 * k])
        k += 1
      tried[c] = k * BATCH / (time.perf_counter() - t0)
    threads = max(tried, key=tried.get)
  oracle.set_num_threads(threads)
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
    "kind": "port", "affinity_cores": affinity, "affinity_cores_tried": affinity <= 128 or threads == affinity,
    "threads_tried": {str(k): v for k, v in tried.items()},
    "sample": f"{n} batches of {BATCH} (float64 C oracle, OpenMP, {dt:.1f} s)",
  }


def timeit(fn, min_seconds=0.5, min_calls=3):
  """median seconds per call over >= min_calls calls and >= min_seconds of wall time."""
  fn(); torch.cuda.synchronize()
  ts, t_all = [], time.perf_counter()
  while len(ts) < min_calls or time.perf_counter() - t_all < min_seconds:
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    if len(ts) >= 2000:
      break
  return float(np.median(ts)), len(ts)


def per_call_section(eng, model, params, noise1, t_slices, y1, lp1, dev):
  """The reference's literal call pattern: one call per 65 536-sample batch."""
  n_calls, S = 200, t_slices.numel()
  out = {}

  def loop(call):
    for _ in range(20):
      call(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_calls):
      call(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n_calls

  dt = loop(lambda i: eng.sample_logprob(noise1, t_slices[i % S:i % S + 1], out=y1, logp_out=lp1))
  out["engine_eager"] = {"ms_per_call": dt * 1e3, "value": BATCH / dt, "kernel": eng.last_path(),
                         "note": "FlowEngine.sample_logprob, preallocated outputs, one launch per batch"}
  cond_b1 = [t_slices[i:i + 1].expand(BATCH).reshape(BATCH, 1).contiguous() for i in range(min(S, 8))]
  dt = loop(lambda i: model.apply.sample_and_log_prob(params, cond=cond_b1[i % len(cond_b1)], noise=noise1,
                                                       sample_shape=(BATCH,)))
  out["model_apply_eager"] = {"ms_per_call": dt * 1e3, "value": BATCH / dt, "kernel": eng.last_path(),
                              "note": "model.apply.sample_and_log_prob(params, cond=[B,1]) -- the reference's literal "
                                      "form (applications.py:153-158), outputs allocated per call, parameters "
                                      "prepared on every call (cnf_model_set_params: pure-function semantics)"}
  model.assume_unchanged_params = True       # opt-in: skip the preparation while torch sees no write to `params`
  dt = loop(lambda i: model.apply.sample_and_log_prob(params, cond=cond_b1[i % len(cond_b1)], noise=noise1,
                                                       sample_shape=(BATCH,)))
  model.assume_unchanged_params = False
  out["model_apply_eager_assume_unchanged"] = {"ms_per_call": dt * 1e3, "value": BATCH / dt,
                                               "note": "the same with FlowModel(assume_unchanged_params=True)"}
  dt = loop(lambda i: model.apply.sample_and_log_prob(params, cond=cond_b1[i % len(cond_b1)], seed=1000 + i,
                                                       sample_shape=(BATCH,)))
  out["model_apply_seeded"] = {"ms_per_call": dt * 1e3, "value": BATCH / dt, "kernel": eng.last_path(),
                               "bytes_per_sample": 4 * (DIM + 1), "hbm_frac": 4 * (DIM + 1) * BATCH / dt / 1e9 / PEAK_HBM_GBS,
                               "note": "model.apply.sample_and_log_prob(params, cond=[B,1], seed=) -- the reference's "
                                       "form with the base draw inside the call (conditional.py:376-402): "
                                       "cnf_sample_logprob_seeded, noise drawn in the flow kernel; algorithmic bytes "
                                       "12 per sample (y and log_prob out, nothing in)"}
  try:       # the same launch-bound loop captured once into a HIP graph and replayed (entry points only enqueue)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
      eng.reserve(1)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
      for i in range(n_calls):
        eng.sample_logprob(noise1, t_slices[i % S:i % S + 1], out=y1, logp_out=lp1)
    graph.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
      graph.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10 / n_calls
    out["hip_graph"] = {"ms_per_call": dt * 1e3, "value": BATCH / dt,
                        "note": f"{n_calls} calls captured in one graph, replayed"}
  except Exception as exc:      # reported, never fatal for the metric
    out["hip_graph"] = {"error": repr(exc)[:200]}
  return out


def dim10_section(dev):
  """sample_and_log_prob at dim 10 against ITS roofline: fp32 ALU, 24.3 kflop per sample."""
  from cnf_ot_amd import FlowConfig, FlowEngine, Params
  cfg = FlowConfig(dim=10)
  eng = FlowEngine(cfg, dev).load(Params.random(cfg, 0.12, seed=42, device=dev))
  out = {}
  for name, n in (("shard_32768", 32768), ("large_2M", 1 << 21)):
    z = eng.normal(42, n)
    c = torch.tensor([0.5], device=dev)
    y, lp = torch.empty_like(z), torch.empty(n, device=dev)
    eng.set_profiling(True)
    sec, calls = timeit(lambda: eng.sample_logprob(z, c, out=y, logp_out=lp), 0.3)
    f_ms, _, launches, smp = eng.read_profile()
    eng.set_profiling(False)
    k = f_ms * 1e-3 / max(launches, 1)
    tf = FLOP_PER_SAMPLE_D10 * n / k / 1e12
    out[name] = {"samples": n, "kernel_ms": k * 1e3, "call_ms": sec * 1e3, "samples_per_s": n / k,
                 "kernel": eng.last_path(),
                 "roofline": {"bound": "alu_fp32", "flop_per_sample": FLOP_PER_SAMPLE_D10, "achieved": tf,
                              "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP32_TFLOPS}}
  return out


def _config_steps(dev, which, world=1):
  """(name, update-step closure, flow passes per step, description) of BASELINE configs 3-5 (per-rank shard
  of the global batch when torch.distributed is initialised: applications._Ctx shards by samples)."""
  from functools import partial
  from cnf_ot_amd import FlowConfig, FlowModel, Params, applications as app, solvers
  if which == "cfg3":
    cfg = FlowConfig(dim=2); B, tb = 131072, 32
    model = FlowModel(cfg); params = Params.random(cfg, 0.2, seed=42, device=dev)
    f = partial(app.rwpo_loss_fn, model, 2, 1.0, 1.0, 0.01, 0.01, tb, "quadratic", 1.0)
    passes = 2 * B + tb * (B // 32) * 7
    desc = "configs[2]: RWPO quadratic T=1 beta=1 dim=2 batch=131072, t_batch_size=32"
  elif which == "cfg4":
    cfg = FlowConfig(dim=10); B, tb = 262144, 32
    model = FlowModel(cfg); params = Params.random(cfg, 0.12, seed=42, device=dev)
    f = partial(app.fp_loss_fn, model, 10, 1.0, 1.0, 0.5, 0.01, 0.01, tb, "ou")
    passes = B + tb * (B // 32) * 23
    desc = "configs[3]: Fokker-Planck OU a=1 sigma=.5 dim=10 batch=262144, t_batch_size=32"
  else:
    cfg = FlowConfig(dim=2); B, tb = 32 << 20, 32     # kinetic + obstacle terms use batch // 32 = 2^20 per slice
    model = FlowModel(cfg); params = Params.random(cfg, 0.2, seed=42, device=dev)
    f = partial(app.ot_loss_fn, model, 2, 1, 0.01, tb, "obstacle", source="gaussian")
    passes = 2 * B + tb * (B // 32) * 3
    desc = "configs[4]: OT obstacle dim=2, 2^20 samples x 32 time-slices (kinetic + obstacle), density fit on 2^25"
  opt = solvers.Adam(1e-3); state = opt.init(params)
  upd = solvers.make_update(f, opt, B)
  k = [0]

  def step():
    k[0] += 1
    return upd(params, 1000 + k[0], 5000.0, state)[0]

  return step, f, params, B, passes, desc


def configs_section(dev):
  out = {}
  from cnf_ot_amd import applications as app
  for which in ("cfg3", "cfg4", "cfg5"):
    try:
      step, f, params, B, passes, desc = _config_steps(dev, which)
      # cfg4 / cfg5 here: ONE GPU's 1/8 share of the global batch (the config is an 8-GPU one)
      share = 8 if which in ("cfg4", "cfg5") else 1
      Bl = B // share
      l_sec, _ = timeit(lambda: f(params, 42, 5000.0, Bl), 0.3)
      vg = app.value_and_grad(f)
      g_sec, _ = timeit(lambda: vg(params, 42, 5000.0, Bl), 0.3)

      def pipelined(fn, n=20):       # n calls enqueued back to back, ONE synchronisation: what a training loop sees
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
          fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

      out[which] = {"workload": desc, "batch_on_this_gpu": Bl, "flow_passes": passes // share,
                    "loss_ms": l_sec * 1e3, "value_and_grad_ms": g_sec * 1e3,
                    "loss_ms_pipelined": pipelined(lambda: f(params, 42, 5000.0, Bl)) * 1e3,
                    "value_and_grad_ms_pipelined": pipelined(lambda: vg(params, 42, 5000.0, Bl)) * 1e3,
                    "timing": "loss_ms / value_and_grad_ms: median of calls each followed by a synchronisation (the "
                              "host's enqueue time is exposed); *_pipelined: 20 calls, one synchronisation",
                    "flow_passes_per_s_loss": passes / share / l_sec}
    except Exception as exc:
      out[which] = {"error": repr(exc)[:300]}
  return out


def _free_port():
  import socket
  with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


def self_launch(args):
  """`python bench.py --gpus N` (N > 1) without a launcher in front: start the N ranks as CHILD processes through
  torch.distributed.run -- the command the driver itself uses -- and hand back their return code.  This process has
  made no GPU call yet (importing torch and counting devices do not initialise the GPU) and makes none: it only
  waits, so nothing that touched the GPU is ever re-exec'ed.  Rank 0 of the children prints the JSON line on the
  inherited stdout."""
  import subprocess
  n_dev = torch.cuda.device_count()
  if not args.share_device and n_dev < args.gpus:
    print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) visible (a one-GPU rehearsal needs "
          f"--backend gloo --share-device)", file=sys.stderr, flush=True)
    return 2
  env = dict(os.environ)
  env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this host driver
  env.setdefault("OMP_NUM_THREADS", "4")
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
  return subprocess.run(cmd, env=env).returncode


def main():
  args = parse_args()
  launched = "WORLD_SIZE" in os.environ
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if not launched and args.gpus > 1:
    sys.exit(self_launch(args))
  if world != args.gpus:
    args.gpus = world
  assert torch.cuda.is_available(), "bench.py needs an MI355X (ROCm) device"
  dev = torch.device("cuda", 0 if args.share_device else local_rank)
  torch.cuda.set_device(dev)
  dist = None
  comm = {"world": world, "backend": None, "launcher": "torch.distributed.run" if launched else "none"}
  if world > 1 or args.force_dist:
    import torch.distributed as dist
    if not launched:                                     # --force-dist, one rank: a rendezvous of its own
      os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
      os.environ.setdefault("MASTER_PORT", str(_free_port()))
      os.environ.setdefault("RANK", "0")
      os.environ.setdefault("WORLD_SIZE", "1")
    if args.backend == "nccl":
      dist.init_process_group("nccl", device_id=dev)
      try:
        comm["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
      except Exception as exc:      # reported, never fatal
        comm["rccl_version"] = repr(exc)[:80]
    else:
      dist.init_process_group(args.backend)
    comm["backend"] = dist.get_backend()
    # one all-reduce through the backend before anything is timed: every rank contributes 1
    chk = torch.ones(1, device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
    dist.all_reduce(chk)
    comm["ranks_seen"] = int(chk.item())
    assert comm["ranks_seen"] == world, "the all-reduce did not see every rank"
  args._comm = comm

  def barrier():
    torch.cuda.synchronize()
    if dist is not None:
      dist.barrier()
    torch.cuda.synchronize()

  def max_over_ranks(x):
    if dist is None:
      return x
    t = torch.tensor([x], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

  if args.workload != "sample":
    return workload_main(args, dev, dist, world, rank, barrier, max_over_ranks)

  from cnf_ot_amd import FlowConfig, FlowEngine, FlowModel, Params
  from cnf_ot_amd.distributed import Shard, shard_range

  # this rank's contiguous block of the step's slices
  s_first, S = shard_range(args.slices, Shard(rank, world))
  cfg = FlowConfig(dim=DIM)
  params = Params.random(cfg, args.param_scale, seed=42, device=dev)
  model = FlowModel(cfg)
  eng = model.engine(dev).load(params)
  eng.reserve(S)                      # table workspace: allocated here, never inside a compute call

  # synthetic inputs, resident in HBM: global slice g is samples [g*BATCH, +BATCH) of the seed-42 Philox stream
  noise = eng.normal(42, S * BATCH, first_sample=s_first * BATCH)
  t_slices = torch.linspace(0.0, 1.0, args.slices, device=dev)[s_first:s_first + S].contiguous()
  y = torch.empty(S * BATCH, DIM, device=dev)
  lp = torch.empty(S * BATCH, device=dev)

  def step():
    if S > 0:
      eng.sample_logprob(noise, t_slices, out=y, logp_out=lp)

  torch.cuda.synchronize()
  t0 = time.perf_counter()
  step()
  torch.cuda.synchronize()
  cold_ms = (time.perf_counter() - t0) * 1e3       # first call: code-object load, LDS attribute; reported, not timed
  for _ in range(max(args.warmup - 1, 0)):
    step()
  barrier()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    step()
  torch.cuda.synchronize()
  elapsed = time.perf_counter() - t0
  if dist is not None:
    dist.barrier()
  elapsed = max_over_ranks(elapsed)
  path = eng.last_path()
  assert torch.isfinite(lp).all(), "non-finite log_prob in the bench output"

  # per-kernel durations of the SAME calls: HIP events recorded by the library on the launch stream around
  # each kernel (cnf_model_set_profiling), over min(steps, 8) further steps -- outside the timed region, so
  # the headline number carries no event overhead
  eng.set_profiling(True)
  torch.cuda.synchronize()
  n_prof = min(args.steps, 8)
  t0 = time.perf_counter()
  for _ in range(n_prof):
    step()
  flow_ms, build_ms, launches, prof_samples = eng.read_profile()
  prof_wall_ms = (time.perf_counter() - t0) * 1e3 / max(n_prof, 1)      # wall time of a PROFILED step (events recorded between the kernels)
  eng.set_profiling(False)
  k_dur = flow_ms * 1e-3 / max(launches, 1)                   # average launch duration of the dominant kernel
  k_samples = prof_samples / max(launches, 1)
  b_dur = build_ms * 1e-3 / max(launches, 1)
  achieved_gbs = BYTES_PER_SAMPLE * k_samples / k_dur / 1e9 if launches else float("nan")
  exec_flop = EXECUTED_FLOP_PER_SAMPLE if path == "tables" else FLOP_PER_SAMPLE
  executed_tflops = exec_flop * k_samples / k_dur / 1e12 if launches else float("nan")
  achieved_tflops = FLOP_PER_SAMPLE * k_samples / k_dur / 1e12 if launches else float("nan")

  total_samples = args.steps * args.slices * BATCH
  value = total_samples / elapsed
  line = {
    "metric": METRIC, "value": value, "unit": "samples/s", "n_gpus": world,
    "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
    "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
    "dtype": "f32", "data": "synthetic",
    "config": {
      "workload": "configs[1]: OT free dim=2, batch=65536, sample_and_log_prob (RQS fwd + log|detJ|), L=2 H=16 M=2 "
                  "K=5, params N(0,%.2f^2) seed 42; ONE STEP = one evaluator-shaped pass of %d time-slices x 65536 "
                  "samples (cnf_ot/utils.py:311-340), each slice its own condition t, one cnf_sample_logprob call "
                  "per rank" % (args.param_scale, args.slices),
      "batch": BATCH, "dim": DIM, "slices_per_step": args.slices, "samples_per_step": args.slices * BATCH,
      "slices_per_step_this_rank": S, "timed_region_s": elapsed,
      "parallelism": f"slice-shard x{world} (contiguous blocks of the step's slices; no collective)",
      "comm": args._comm,
    },
    "cold_first_call_ms": cold_ms,
    "roofline": {
      # HBM: SURVEY.md 8(d)'s algorithmic bytes (20 B per sample: x in, y out, log_prob out) -- the one bound a
      # reformulation cannot move.  The ALU views are beside it.
      "bound": "hbm", "achieved": achieved_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
      "frac": achieved_gbs / PEAK_HBM_GBS, "traffic": None,
      "kernel": KERNEL_NAMES.get(path, path), "path": path,
      "launch_ms": k_dur * 1e3, "launches_timed": launches,
      "samples_per_launch": k_samples, "bytes_per_sample": BYTES_PER_SAMPLE,
      "table_build_ms_per_launch": b_dur * 1e3,
      # consistency: (launch_ms + table_build_ms_per_launch) x launches per step <= the wall time of a PROFILED step;
      # a profiled step is a few per cent slower than a timed one (an event record between every two kernels)
      "profiled_step_ms": prof_wall_ms, "launches_per_step": launches / max(n_prof, 1),
      "timing": "HIP events recorded by the library on the launch stream around each kernel launch "
                "(cnf_model_set_profiling, include/cnf_ot_amd_debug.h), same calls as the timed region, run right after it",
      "alu_executed": {"flop_per_sample": exec_flop, "achieved": executed_tflops,
                       "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": executed_tflops / PEAK_FP32_TFLOPS},
      "alu_reference_formulation": {"flop_per_sample": FLOP_PER_SAMPLE, "achieved": achieved_tflops,
                                    "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                                    "frac": achieved_tflops / PEAK_FP32_TFLOPS},
      "note": "SURVEY.md 8(d) derived an fp32-ALU bound of 60 G samples/s from the reference formulation (2576 flop "
              "per sample: 2-16-16-16 MLP conditioner + splines).  The table path reads the same conditioner from "
              "exact piecewise-linear tables and executes ~448 flop per sample, so that figure is no longer a bound "
              "(alu_reference_formulation.frac can exceed 1); the bound left is HBM.  What limits the kernel is "
              "VALU issue (DESIGN.md 5.1d): ~545 vector instructions per 128-sample wave-tile, 96 of them "
              "quarter-rate transcendentals -> an issue ceiling of ~102 G samples/s for this instruction stream; "
              "0.22 of the HBM roofline is ~0.85 of that ceiling.",
    },
  }
  pmc = os.path.join(ROOT, "profiles", "hbm_traffic.json")
  if os.path.exists(pmc) and launches:
    with open(pmc) as f:
      rec = json.load(f)
    if rec.get("samples_per_launch"):
      # PMC counters (separate rocprofv3 --pmc passes of this command) are per launch of the profiled run;
      # rescaled by samples only when the profiled launch had another size (the kernel streams: bytes ~ samples)
      line["roofline"]["traffic"] = rec["bytes_per_launch"] * k_samples / rec["samples_per_launch"]
      line["roofline"]["traffic_source"] = rec.get("source", "profiles/hbm_traffic.json")
      line["roofline"]["traffic_samples_per_launch"] = rec["samples_per_launch"]

  def extra(name, fn):        # the extra sections are reported, never fatal for the metric
    try:
      line[name] = fn()
    except Exception as exc:
      line[name] = {"error": repr(exc)[:300]}

  if rank == 0 and world == 1 and not args.no_extras:
    nb = min(S, 64)
    extra("per_call", lambda: per_call_section(eng, model, params, noise[:BATCH], t_slices[:nb], y[:BATCH],
                                               lp[:BATCH], dev))
    del noise, y, lp
    torch.cuda.empty_cache()
    extra("dim10", lambda: dim10_section(dev))
    extra("configs", lambda: configs_section(dev))

  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    extra("cpu_baseline", lambda: cpu_baseline(params.flat.cpu().double().numpy(), args.cpu_seconds, args.cpu_threads))
  elif rank == 0:
    line["cpu_baseline"] = None

  if rank == 0:
    print(json.dumps(line), flush=True)
  if dist is not None:
    dist.destroy_process_group()


def workload_main(args, dev, dist, world, rank, barrier, max_over_ranks):
  """--workload cfg3|cfg4|cfg5: a step = update() = value_and_grad + Adam of that BASELINE config, global batch
  sample-sharded over the ranks, ONE sum all-reduce of [partial sums, gradient] per step (RCCL through
  torch.distributed; applications._Ctx.reduce)."""
  step, f, params, B, passes, desc = _config_steps(dev, args.workload, world)
  from cnf_ot_amd import applications as app

  def timed():
    for _ in range(max(args.warmup, 1)):
      loss = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
      loss = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist is not None:
      dist.barrier()
    return max_over_ranks(el), loss

  modes = None
  if args.workload == "cfg5" and world > 1:
    # configs[4] "allreduce/compute overlap": the density-fit terms' all-reduce under the kinetic / obstacle slices
    # (shipped default) against ONE blocking collective at the end -- both timed, the same way
    app.OVERLAP_ALLREDUCE = False
    el_block, _ = timed()
    app.OVERLAP_ALLREDUCE = True
    elapsed, loss = timed()
    modes = {"overlapped_ms_per_step": elapsed / args.steps * 1e3, "blocking_ms_per_step": el_block / args.steps * 1e3,
             "shipped": "overlapped"}
  else:
    elapsed, loss = timed()
  assert torch.isfinite(torch.as_tensor(loss)).all(), "non-finite loss"
  line = {
    "metric": "flow passes/sec through update() (value_and_grad + Adam) of " + args.workload,
    "value": passes * args.steps / elapsed, "unit": "flow passes/s", "n_gpus": world, "steps": args.steps,
    "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
    "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
    "config": {"workload": desc, "global_batch": B, "flow_passes_per_step": passes,
               "parallelism": f"sample-shard x{world}; one sum all-reduce of partial sums + gradient "
                              f"({params.flat.numel()} floats) per step",
               "comm": args._comm},
    "loss": float(loss),
  }
  if modes:
    line["allreduce"] = modes
  if rank == 0:
    print(json.dumps(line), flush=True)
  if dist is not None:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
