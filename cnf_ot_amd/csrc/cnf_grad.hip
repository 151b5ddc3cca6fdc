// cnf_grad.hip -- backward pass of the fused Monte-Carlo loss terms and the
// Adam update: the MI355X replacement of `jax.value_and_grad(loss_fn)` +
// `optax.adam` in cnf_ot/mfc/solvers.py:90-97.
//
// One launch = one loss term over (time-slice, sample tile), like loss_kernel,
// but it also back-propagates the term through every flow pass it made:
//   * each pass is re-run with its layer inputs stashed in LDS (recompute
//     instead of storing activations), then differentiated layer by layer in
//     reverse: spline partials (cnf_backward.h) -> conditioner backward;
//   * data backprop through the 16x16 layers uses scalar (SGPR) weights;
//   * weight gradients are batch GEMMs on the matrix cores
//     (v_mfma_f32_16x16x4_f32), accumulated straight into a per-wave gradient
//     slab in global memory (plain read-modify-write: a slab has one owner, so
//     the result is deterministic); grad_finish_kernel sums the slabs.
// Built for the reference's network only: hidden 16, 2 hidden layers, 5 bins.
#include "cnf_backward.h"
#include "cnf_pwl.h"

#include <math.h>

namespace cnf {

constexpr int GK = 5;              // bins
constexpr int GP = 3 * GK + 1;     // 16 spline parameters
// One sample per lane; a tile = one workgroup's samples = blockDim.x (64, 128 or 256: the host picks the size that
// puts the most waves on a CU -- a tile's LDS working set is (L + 3 .. L + 9) x D floats per sample plus 8.7 KB of
// MFMA staging per wave, and at 1 wave per SIMD a wave issues at half rate with every latency exposed).
#define GTS ((int)blockDim.x)
constexpr int GTS_MAX = TILE;
// LDS geometry of a tile.  Every [D][samples] buffer (noise, stashes, adjoints ...) has rows of GROW = GTS + 4 floats:
// a wave's 64 samples of one input dimension are contiguous, and rows read 16 bytes per lane along the samples (the
// A operand of the first layer's weight-gradient GEMM, cnf_backward.h) fall into different banks.  In front of them:
// the `first` spline's table, 64 ones (the bias row of the first layer's MFMA operands) and the tile's condition column.
#define GROW (GTS + 4)
constexpr int HDRG = hdr_floats(GK);
constexpr int ONES_OFF = HDRG, C_OFF = HDRG + 64;
#define PRE_FLOATS (HDRG + 64 + GROW)

__device__ __forceinline__ void tile_consts(float* lds) {
  for (int i = threadIdx.x; i < 64; i += GTS) lds[ONES_OFF + i] = 1.0f;
}

// One launch evaluates up to GRAD_MAX_JOBS loss terms ("jobs": the terms of one composite loss, each with its own
// points, slices and coefficient): their tiles share the grid, so small terms run side by side instead of one
// under-filled launch after the other (a default-config training step is three launches of 8 tiles each).
constexpr int GRAD_MAX_JOBS = 4;
struct GradJob {
  CnfLossSpec spec;
  const float* pts;
  const float* t;
  double* sums;          // [n_slices] loss-term sums (value of value_and_grad)
  int64_t B, n_slices, pts_slice_stride;
  int64_t first_tile;    // of this job, in the launch's tile numbering
  float scale;           // d(total loss) / d(this term's sum)
};
struct GradArgs {
  ModelArgs m;
  GradJob job[GRAD_MAX_JOBS];
  int32_t n_jobs;
  float* slabs;          // [gridDim.x * 4][n_params]
  int64_t n_params;
  int64_t n_tiles;       // of all jobs
  uint32_t div_magic;
};

// A wave zeroes its own gradient slab before its first tile (the host used to clear all slabs with one memset
// per term: a 10 MB fill kernel and its launch in front of every backward launch).  Same wave, same addresses
// later: the read-modify-writes that follow are ordered behind these stores.
__device__ __forceinline__ void slab_clear(float* gslab, int64_t n_params, int lane) {
  for (int64_t j = lane; j < n_params; j += 64) gslab[j] = 0.0f;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}

struct FirstAcc { float Wb[GK], Hb[GK], Db[GK + 1]; };

__device__ __forceinline__ void tile_load1(const float* __restrict__ g, float* U, int D, uint32_t magic,
                                           int64_t tile_start, int64_t B) {
  const int64_t base = tile_start * D;
  const int n_el = (int)(B - tile_start < GTS ? B - tile_start : GTS) * D;
  for (int e = threadIdx.x; e < GTS * D; e += GTS) {
    const int s = magic ? (int)__umulhi((uint32_t)e, magic) : e, d = e - s * D;
    U[d * GROW + s] = e < n_el ? g[base + e] : 0.0f;
  }
}

// Offsets inside a flow layer's conditioners in 32-bit closed form (H = 16, M = 2, P = 16): conditioner d has
// cond_floats(d) = 16 d + 576 flat weights and cond_floats_mfma(d, 2) = 256 (d + 6) MFMA-layout floats (64-bit loops
// over d here cost ~130 scalar instructions per layer and a dozen scalar registers spilled to vector lanes)
__device__ __forceinline__ int cond_size(int d) { return 16 * d + 576; }
__device__ __forceinline__ int cond_size_q(int d) { return 256 * (d + 6); }
__device__ __forceinline__ int cond_prefix(int d) { return (d - 1) * (8 * d + 576); }          // conditioners 1 .. d-1
__device__ __forceinline__ int cond_prefix_q(int d) { return (d - 1) * (128 * d + 1536); }
static_assert(cond_floats(3, 16, 2, 16) == 16 * 3 + 576 && cond_floats_mfma(3, 2) == 256 * (3 + 6), "conditioner sizes");

// forward of one pass with every layer input kept: St[s] is the input of step s
// (St[0] filled by the caller), St[L] the result.  Returns the log-det sum.
// The direction is a RUNTIME (wave-uniform) flag and every pass of the kernel
// goes through the single call site of this function and of pass_bwd: the
// conditioner code (the bulk) exists once, so the kernel fits the instruction
// cache (the first version inlined 10 forward + 6 backward copies: 34 k
// instructions, 4x the cache).
template <bool FAST, int DFIX = 0>
__device__ __forceinline__ float pass_fwd_stash(const ModelArgs& a, float* lds, float* St, float c, bool to_base) {
  const int D = DFIX ? DFIX : a.D;
  const float* tab = lds;
  const int wbase = threadIdx.x & ~63;
  lds[C_OFF + threadIdx.x] = c;
  uniform_ptr weights = as_uniform(a.prep + hdr_floats(GK));
  float acc = 0.0f;
  for (int s = 0; s < a.L; ++s) {
    const int l = to_base ? a.L - 1 - s : s;
    const bool odd = l & 1;
    const int first_idx = odd ? D - 1 : 0, idx_step = odd ? -1 : 1;
    const float* cu = St + s * D * GROW + threadIdx.x;
    float* co = St + (s + 1) * D * GROW + threadIdx.x;
    uniform_ptr w = weights + l * (int)a.per_layer;
    [[maybe_unused]] const float* wflat = a.prep + hdr_floats(GK) + l * (int)a.per_layer;
    [[maybe_unused]] const float* wq = a.wq + l * (int)a.per_layer_q;
    [[maybe_unused]] int qo = 0;       // this conditioner's offset in the layer's MFMA-layout weights
    [[maybe_unused]] CondW cw;         // the NEXT conditioner's weights: one conditioner ahead (cnf_backward.h)
    if constexpr (FAST) { if (D > 1) cw = cond_weights(wq, 0, wflat, 1); __builtin_amdgcn_sched_barrier(0); }
    float o, ld;
    if (to_base) table_spline<GK, false, FAST, float>(tab, cu[first_idx * GROW], a.sc, o, ld);
    else table_spline<GK, true, FAST, float>(tab, cu[first_idx * GROW], a.sc, o, ld);
    co[first_idx * GROW] = o;
    acc += ld;
    for (int d = 1; d < D; ++d) {
      const int i = first_idx + d * idx_step;
      float th[GP];
      if constexpr (FAST) {        // matrix cores (cnf_backward.h); `wq` walks the MFMA-layout weights
        const CondGeom G{(int)((to_base ? co : cu) - lds) - (int)threadIdx.x + wbase, C_OFF + wbase, ONES_OFF, first_idx,
                         idx_step, GROW, d};
        const CondW cur = cw;
        qo += cond_size_q(d);
        wflat += cond_size(d);
        if (d + 1 < D) cw = cond_weights(wq, qo, wflat, d + 1);
        __builtin_amdgcn_sched_barrier(0);
        float h1m[4][4], h2m[4][4];
        cond_fwd_mfma(lds, G, wq, cur, h1m, h2m, th);
      } else {
        conditioner<16, GP, float>(w, d, 2, c, to_base ? co : cu, first_idx, idx_step, GROW, th);
        w += cond_size(d);
      }
      if (to_base) cond_spline<GK, false, FAST, float>(th, cu[i * GROW], a.sc, o, ld);
      else cond_spline<GK, true, FAST, float>(th, cu[i * GROW], a.sc, o, ld);
      co[i * GROW] = o;
      acc += ld;
    }
  }
  return acc;
}

// backward of the pass whose stash is in St.  Aa holds the adjoint of the final
// output on entry; the function ping-pongs between Aa and Ab and returns the
// buffer that holds the adjoint of the pass input.
// fa_lds (optional): the `first` spline's per-bin adjoint sums live in 16 LDS rows instead of `fa` -- sixteen registers
// that are then free during the conditioner loop (the generic-dimension loss kernel has none to spare)
// AHEAD: the recomputation weights of the next conditioner are fetched during the current one's backward (twelve
// registers held across it; the generic-dimension loss kernel, whose tile state is larger, fetches them at use)
template <bool FAST, bool WGRAD = true, int DFIX = 0, bool AHEAD = true>
__device__ __forceinline__ float* pass_bwd(const ModelArgs& a, float* lds, const float* St, float* Aa,
                                           float* Ab, float ld_bar, float c, bool to_base, float* gslab,
                                           float* stage, FirstAcc& fa, float* fa_lds = nullptr) {
  const int D = DFIX ? DFIX : a.D;
  const float* tab = lds;
  const int wbase = threadIdx.x & ~63;
  lds[C_OFF + threadIdx.x] = c;
  uniform_ptr weights = as_uniform(a.prep + hdr_floats(GK));
  float* Aout = Aa;
  float* Ain = Ab;
  for (int s = a.L - 1; s >= 0; --s) {
    const int l = to_base ? a.L - 1 - s : s;
    const bool odd = l & 1;
    const int first_idx = odd ? D - 1 : 0, idx_step = odd ? -1 : 1;
    const float* cu = St + s * D * GROW + threadIdx.x;
    const float* co = St + (s + 1) * D * GROW + threadIdx.x;
    float* ao = Aout + threadIdx.x;
    float* au = Ain + threadIdx.x;
    for (int d = 0; d < D; ++d) au[d * GROW] = 0.0f;
    const int lay = l * (int)a.per_layer, layq = l * (int)a.per_layer_q;
    int off = cond_prefix(D);                          // end of this layer's conditioners
    [[maybe_unused]] int offq = cond_prefix_q(D);      // the same in the MFMA-layout weights
    [[maybe_unused]] CondW cw;                         // the NEXT conditioner's recomputation weights, one conditioner ahead
    if constexpr (FAST && AHEAD) {
      if (D > 1) {
        cw = cond_weights(a.wq + layq, offq - cond_size_q(D - 1), a.prep + hdr_floats(GK) + lay + off - cond_size(D - 1), D - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    for (int d = D - 1; d >= 1; --d) {
      off -= cond_size(d);
      const int i = first_idx + d * idx_step;
      float* gw = WGRAD ? gslab + GP + lay + off : nullptr;
      WgradPre pre;
      float th[GP], tb[GP];
      if constexpr (FAST) {      // recompute, data backprop and weight gradients on the matrix cores (cnf_backward.h)
        offq -= cond_size_q(d);
        const float* wflat = a.prep + hdr_floats(GK) + lay + off;
        const CondGeom G{(int)((to_base ? co : cu) - lds) - (int)threadIdx.x + wbase, C_OFF + wbase, ONES_OFF, first_idx,
                         idx_step, GROW, d};
        float h1m[4][4], h2m[4][4];
        if constexpr (!AHEAD) cw = cond_weights(a.wq + layq, offq, wflat, d);
        cond_fwd_mfma(lds, G, a.wq + layq, cw, h1m, h2m, th);
        // fetched here, the spline backward ahead of their first use: the first data-backprop operand and the first
        // accumulator tile of the weight gradient (the others at the start of cond_bwd_mfma)
        const int o_wo = (1 + d) * 16 + 16 + 256 + 16;
        const f4 Ao = cond_weight_T(wflat + o_wo);
        WgradAcc pre_o;
        if constexpr (WGRAD) pre_o = wgrad_fetch(gw + o_wo, 16, gw + o_wo + 256);
        __builtin_amdgcn_sched_barrier(0);
        float vb;
        if (to_base) vb = cond_spline_bwd<GK, false, FAST>(th, cu[i * GROW], co[i * GROW], ao[i * GROW], ld_bar, a.sc, tb);
        // (base -> data: the output is formed again inside, in the bin the backward selects -- next to a knot the stashed
        // one can lie a rounding outside that bin, and 1 / f' of a clipped position is garbage: soak_vjp_case_1_16.log)
        else vb = cond_spline_bwd<GK, true, FAST>(th, cu[i * GROW], co[i * GROW], ao[i * GROW], ld_bar, a.sc, tb, nullptr, nullptr,
                                                  nullptr, true);
        au[i * GROW] += vb;
        // the next conditioner's recomputation weights: covered by this one's backward
        if constexpr (AHEAD) { if (d > 1) cw = cond_weights(a.wq + layq, offq - cond_size_q(d - 1), wflat - cond_size(d - 1), d - 1); }
        cond_bwd_mfma<WGRAD>(lds, G, (int)((to_base ? Aout : Ain) - lds) + wbase, wflat, Ao, h1m, h2m, tb, gw, stage, pre_o);
      } else {
        uniform_ptr w = weights + lay + off;
        if constexpr (WGRAD) pre = wgrad_prefetch(gw, d);
        float h1[16], h2[16];
        conditioner_keep(w, d, c, to_base ? co : cu, first_idx, idx_step, GROW, h1, h2, th);
        float vb;
        if (to_base) vb = cond_spline_bwd<GK, false, FAST>(th, cu[i * GROW], co[i * GROW], ao[i * GROW], ld_bar, a.sc, tb);
        else vb = cond_spline_bwd<GK, true, FAST>(th, cu[i * GROW], co[i * GROW], ao[i * GROW], ld_bar, a.sc, tb, nullptr, nullptr,
                                                  nullptr, true);
        au[i * GROW] += vb;
        conditioner_bwd<WGRAD>(w, d, c, to_base ? co : cu, first_idx, idx_step, GROW, h1, h2, tb, to_base ? ao : au,
                               gw, stage, pre);
      }
    }
    float vb0;
    if (fa_lds) {
      FirstAcc t0;
#pragma unroll
      for (int j = 0; j < GK; ++j) { t0.Wb[j] = 0.0f; t0.Hb[j] = 0.0f; }
#pragma unroll
      for (int j = 0; j <= GK; ++j) t0.Db[j] = 0.0f;
      if (to_base) vb0 = table_spline_bwd<GK, false, FAST>(tab, cu[first_idx * GROW], co[first_idx * GROW], ao[first_idx * GROW],
                                                           ld_bar, a.sc, t0.Wb, t0.Hb, t0.Db);
      else vb0 = table_spline_bwd<GK, true, FAST>(tab, cu[first_idx * GROW], co[first_idx * GROW], ao[first_idx * GROW],
                                                  ld_bar, a.sc, t0.Wb, t0.Hb, t0.Db);
      float* f = fa_lds + threadIdx.x;
#pragma unroll
      for (int j = 0; j < GK; ++j) { f[j * GROW] += t0.Wb[j]; f[(GK + j) * GROW] += t0.Hb[j]; }
#pragma unroll
      for (int j = 0; j <= GK; ++j) f[(2 * GK + j) * GROW] += t0.Db[j];
    } else if (to_base) {
      vb0 = table_spline_bwd<GK, false, FAST>(tab, cu[first_idx * GROW], co[first_idx * GROW], ao[first_idx * GROW],
                                              ld_bar, a.sc, fa.Wb, fa.Hb, fa.Db);
    } else {
      vb0 = table_spline_bwd<GK, true, FAST>(tab, cu[first_idx * GROW], co[first_idx * GROW], ao[first_idx * GROW],
                                             ld_bar, a.sc, fa.Wb, fa.Hb, fa.Db);
    }
    au[first_idx * GROW] += vb0;
    float* t = Aout; Aout = Ain; Ain = t;
  }
  return Aout;
}

__device__ __forceinline__ float base_lp(const float* col, int D) {
  float b = 0.0f;
  for (int d = 0; d < D; ++d) { const float x = col[d * GROW]; b = fmaf(-0.5f * x, x, b); }
  return b - (float)(D * HALF_LOG_2PI);
}

// R3b[e] -= sum_d ubar_d * d drift_d / d r_e   (flow_matching_loss_fn's target field)
__device__ __forceinline__ void drift_vjp(const float* r3, const float* ub, float* r3b, int D, int subtype, float a) {
  switch (subtype) {
    case CNF_DRIFT_SMILE: {
      const float x = r3[0], y = r3[GROW], q = x * x + y * y - 4.0f, u0 = ub[0], u1 = ub[GROW];
      r3b[0] -= u0 * (-a * (q + 2.0f * x * x)) + u1 * (-a * 2.0f * x * y);
      r3b[GROW] -= u0 * (-a * 2.0f * x * y) + u1 * (-a * (q + 2.0f * y * y + 2.0f));
      break;
    }
    case CNF_DRIFT_NONGRADIENT: {
      const float u0 = ub[0], u1 = ub[GROW];
      r3b[0] -= u0 * (-a) + u1 * 0.5f;
      r3b[GROW] -= u0 * (-0.5f) + u1 * (-a);
      break;
    }
    case CNF_DRIFT_LORENZ: {
      const float x = r3[0], y = r3[GROW], z = r3[2 * GROW], u0 = ub[0], u1 = ub[GROW], u2 = ub[2 * GROW];
      r3b[0] -= u0 * -10.0f + u1 * (28.0f - 9.0f * z) + u2 * 9.0f * y;
      r3b[GROW] -= u0 * 10.0f - u1 + u2 * 9.0f * x;
      r3b[2 * GROW] -= u1 * (-9.0f * x) + u2 * (-8.0f / 3.0f);
      break;
    }
    default:
      for (int d = 0; d < D; ++d) r3b[d * GROW] += a * ub[d * GROW];
  }
}

// Roles of the steps of a tile's pass program.  Every step is: build the pass
// input, run the forward with stash, act on its result, optionally seed and run
// the backward, act on the input adjoint.
enum Role {
  R_NEG,          // -log_prob of the points (data -> base)
  R_POT, R_RKL,   // potential / reverse KL (base -> data)
  R_R1, R_R2,     // finite-difference velocity: r1 kept, then velocity formed
  R_R3,           // sample at t for the score terms
  R_LPP, R_LPM,   // log_prob at r3 +- dx/2 e_d (forward only / forward + backward of the minus pass)
  R_LPPB,         // the plus pass again, with backward
  R_R3B, R_R2B, R_R1B   // backward of r3 / r2 / r1 (forward recomputed)
};

// DFIX > 0: the event dimension as a compile-time constant (dim 2: the reference's main configurations)
template <bool FAST, int DFIX = 0>
__global__ __launch_bounds__(GTS_MAX, 2) void grad_kernel(const GradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int HDR = hdr_floats(GK);
  const int D = DFIX ? DFIX : a.m.D, L = a.m.L;
  const int DT = D * GROW;
  float* Nn = lds + PRE_FLOATS;
  float* St = Nn + DT;                  // (L+1) stashes
  float* Aa = St + (L + 1) * DT;
  float* Ab = Aa + DT;
  float* V = Ab + DT;
  float* R3 = V + DT;
  float* R3b = R3 + DT;
  float* Ub = R3b + DT;
  float* stage = Ub + DT + (threadIdx.x >> 6) * STAGE_FLOATS;
  // generic dimension: the `first` spline's 16 accumulators per lane in LDS (pass_bwd), not in registers
  float* fa_lds = DFIX ? nullptr : Ub + DT + (GTS >> 6) * STAGE_FLOATS;
  for (int i = threadIdx.x; i < HDR; i += GTS) lds[i] = a.m.prep[i];
  tile_consts(lds);
  if (!DFIX) { for (int j = 0; j < GP; ++j) fa_lds[j * GROW + threadIdx.x] = 0.0f; }
  const int tid = threadIdx.x;
  // (wave-uniform: a scalar register pair, not two vector registers)
  float* gslab = a.slabs + (int64_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (GTS >> 6) + (tid >> 6))) * a.n_params;
  slab_clear(gslab, a.n_params, tid & 63);
  FirstAcc fa;
#pragma unroll
  for (int j = 0; j < GK; ++j) { fa.Wb[j] = 0.0f; fa.Hb[j] = 0.0f; }
#pragma unroll
  for (int j = 0; j <= GK; ++j) fa.Db[j] = 0.0f;

  SliceSum ssum;
  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    int jb = 0;
#pragma unroll
    for (int q = 1; q < GRAD_MAX_JOBS; ++q) jb += (q < a.n_jobs && tile >= a.job[q].first_tile) ? 1 : 0;
    const GradJob& J = a.job[jb];
    const int kind = J.spec.kind;
    const float dt = J.spec.dt, dx = J.spec.dx, coef = J.spec.coef;
    const bool score = kind == CNF_TERM_KINETIC_SCORE || kind == CNF_TERM_FLOW_MATCHING;
    // number of steps of the pass program
    const int n_steps = kind == CNF_TERM_KINETIC ? 3 : (score ? 3 + 3 * D + 3 : 1);
    const int64_t tiles_per_slice = (J.B + GTS - 1) / GTS;
    const int64_t lt = tile - J.first_tile;
    const int64_t slice = lt / tiles_per_slice;
    const int64_t tile_start = (lt - slice * tiles_per_slice) * GTS;
    const bool valid = tile_start + tid < J.B;
    const float sc = valid ? J.scale : 0.0f;
    __syncthreads();
    tile_load1(J.pts + slice * J.pts_slice_stride * D, Nn, D, a.div_magic, tile_start, J.B);
    const float t = J.t[slice];
    __syncthreads();
    float* n_ = Nn + tid;
    float* s0 = St + tid;               // stash 0 column
    float* sL = St + L * DT + tid;      // final output column
    float* aa = Aa + tid;
    float* v_ = V + tid;
    float* r3 = R3 + tid;
    float* r3b = R3b + tid;
    float* ub = Ub + tid;
    float lossv = 0.0f, lp_plus = 0.0f, ubar = 0.0f;

    for (int st = 0; st < n_steps; ++st) {
      // ---- decode the step
      int role, dd = 0;
      if (kind == CNF_TERM_NEG_LOGPROB) role = R_NEG;
      else if (kind == CNF_TERM_POTENTIAL) role = R_POT;
      else if (kind == CNF_TERM_REVERSE_KL) role = R_RKL;
      else if (kind == CNF_TERM_KINETIC) role = st == 0 ? R_R1 : (st == 1 ? R_R2 : R_R1B);
      else {
        if (st < 3) role = st == 0 ? R_R1 : (st == 1 ? R_R2 : R_R3);
        else if (st < 3 + 3 * D) { dd = (st - 3) / 3; const int q = (st - 3) - 3 * dd; role = q == 0 ? R_LPP : (q == 1 ? R_LPM : R_LPPB); }
        else { const int q = st - (3 + 3 * D); role = q == 0 ? R_R3B : (q == 1 ? R_R2B : R_R1B); }
      }
      const bool to_base = role == R_NEG || role == R_LPP || role == R_LPM || role == R_LPPB;
      const float c = (role == R_R1 || role == R_R1B) ? t - 0.5f * dt
                    : ((role == R_R2 || role == R_R2B) ? t + 0.5f * dt : t);
      // ---- pass input
      if (role == R_LPP || role == R_LPM || role == R_LPPB) {
        for (int e = 0; e < D; ++e) s0[e * GROW] = r3[e * GROW];
        s0[dd * GROW] += role == R_LPM ? -0.5f * dx : 0.5f * dx;
      } else {
        for (int e = 0; e < D; ++e) s0[e * GROW] = n_[e * GROW];
      }
      const float ldsum = pass_fwd_stash<FAST, DFIX>(a.m, lds, St, c, to_base);
      // ---- act on the result, prepare the seeds
      bool do_bwd = false;
      float ld_bar = 0.0f;
      switch (role) {
        case R_NEG: {
          lossv = -(base_lp(sL, D) + ldsum);
          for (int e = 0; e < D; ++e) aa[e * GROW] = sc * sL[e * GROW];      // d(-lp)/dx_e = x_e
          ld_bar = -sc; do_bwd = true;
          break;
        }
        case R_POT: {
          const float pa = J.spec.a;
          if (J.spec.subtype == CNF_POT_DOUBLE_WELL) {
            float sm = 0.0f, sp = 0.0f;
            for (int e = 0; e < D; ++e) { const float r = sL[e * GROW]; sm = fmaf(r - pa, r - pa, sm); sp = fmaf(r + pa, r + pa, sp); }
            lossv = sm * sp * 0.25f;
            for (int e = 0; e < D; ++e) { const float r = sL[e * GROW]; aa[e * GROW] = sc * 0.5f * ((r - pa) * sp + (r + pa) * sm); }
          } else {
            float s2 = 0.0f;
            for (int e = 0; e < D; ++e) { const float r = sL[e * GROW]; s2 = fmaf(r, r, s2); }
            if (J.spec.subtype == CNF_POT_OBSTACLE) {
              lossv = 50.0f * expf(-0.5f * s2);
              for (int e = 0; e < D; ++e) aa[e * GROW] = -sc * lossv * sL[e * GROW];
            } else {
              lossv = 0.5f * s2;
              for (int e = 0; e < D; ++e) aa[e * GROW] = sc * sL[e * GROW];
            }
          }
          do_bwd = true;
          break;
        }
        case R_RKL: {
          const float lp = base_lp(n_, D) - ldsum;
          float s2 = 0.0f;
          for (int e = 0; e < D; ++e) { const float r = sL[e * GROW]; s2 = fmaf(r, r, s2); }
          const float Tt = J.spec.T, vs = 2.0f / J.spec.beta * (Tt + 1.0f), vt = 2.0f / J.spec.beta;
          const float ws = (Tt - t) / Tt, wt = t / Tt;
          const float ls = -0.5f * D * logf(6.283185307179586f * vs), lt = -0.5f * D * logf(6.283185307179586f * vt);
          const float as = -0.5f * s2 / vs + ls, at = -0.5f * s2 / vt + lt;
          const float mx = fmaxf(as, at);
          const float es = expf(as - mx) * ws, et = expf(at - mx) * wt;
          lossv = lp - (mx + logf(es + et));
          const float g = (es / vs + et / vt) / (es + et);      // -d logmix / d y_e = g * y_e
          for (int e = 0; e < D; ++e) aa[e * GROW] = sc * g * sL[e * GROW];
          ld_bar = -sc; do_bwd = true;
          break;
        }
        case R_R1:
          for (int e = 0; e < D; ++e) v_[e * GROW] = sL[e * GROW];
          break;
        case R_R2: {
          const float inv_dt = 1.0f / dt;
          for (int e = 0; e < D; ++e) v_[e * GROW] = (sL[e * GROW] - v_[e * GROW]) * inv_dt;      // velocity
          if (kind == CNF_TERM_KINETIC) {
            for (int e = 0; e < D; ++e) {
              const float v = v_[e * GROW];
              lossv = fmaf(v, v, lossv);
              ub[e * GROW] = 2.0f * sc * v;
              aa[e * GROW] = ub[e * GROW] * inv_dt;
            }
            do_bwd = true;                          // the r2 stash is live
          }
          break;
        }
        case R_R3:
          for (int e = 0; e < D; ++e) { r3[e * GROW] = sL[e * GROW]; r3b[e * GROW] = 0.0f; }
          break;
        case R_LPP:
          lp_plus = base_lp(sL, D) + ldsum;
          break;
        case R_LPM: {
          const float lp_minus = base_lp(sL, D) + ldsum;
          float u = fmaf((lp_plus - lp_minus) / dx, coef, v_[dd * GROW]);
          if (kind == CNF_TERM_FLOW_MATCHING) u -= drift_of<float>(r3, dd, D, GROW, J.spec.subtype, J.spec.a);
          lossv = fmaf(u, u, lossv);
          ubar = 2.0f * sc * u;
          ub[dd * GROW] = ubar;
          ld_bar = -ubar * coef / dx;                // d u / d lp_minus
          for (int e = 0; e < D; ++e) aa[e * GROW] = -ld_bar * sL[e * GROW];     // d base / d x_e = -x_e
          do_bwd = true;
          break;
        }
        case R_LPPB:
          ld_bar = ubar * coef / dx;
          for (int e = 0; e < D; ++e) aa[e * GROW] = -ld_bar * sL[e * GROW];
          do_bwd = true;
          break;
        case R_R3B:
          if (kind == CNF_TERM_FLOW_MATCHING) drift_vjp(r3, ub, r3b, D, J.spec.subtype, J.spec.a);
          for (int e = 0; e < D; ++e) aa[e * GROW] = r3b[e * GROW];
          do_bwd = true;
          break;
        case R_R2B:
          for (int e = 0; e < D; ++e) aa[e * GROW] = ub[e * GROW] / dt;
          do_bwd = true;
          break;
        default:   // R_R1B
          for (int e = 0; e < D; ++e) aa[e * GROW] = -ub[e * GROW] / dt;
          do_bwd = true;
          break;
      }
      if (do_bwd) {
        const float* ain = pass_bwd<FAST, true, DFIX, DFIX != 0>(a.m, lds, St, Aa, Ab, ld_bar, c, to_base, gslab, stage, fa, fa_lds) + tid;
        if (role == R_LPM || role == R_LPPB)
          for (int e = 0; e < D; ++e) r3b[e * GROW] += ain[e * GROW];
      }
    }
    float part = valid ? lossv : 0.0f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    ssum.add(J.sums, slice, part);
  }
  ssum.flush();
  // per-bin adjoint sums of the shared `first` spline: wave reduce, one owner write
  float red[GP];
  if (!DFIX) {
#pragma unroll
    for (int j = 0; j < GP; ++j) red[j] = fa_lds[j * GROW + tid];
  } else {
#pragma unroll
    for (int j = 0; j < GK; ++j) { red[j] = fa.Wb[j]; red[GK + j] = fa.Hb[j]; }
#pragma unroll
    for (int j = 0; j <= GK; ++j) red[2 * GK + j] = fa.Db[j];
  }
#pragma unroll
  for (int j = 0; j < GP; ++j) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) red[j] += __shfl_xor(red[j], off, 64);
  }
  if ((tid & 63) == 0) {
#pragma unroll
    for (int j = 0; j < GP; ++j) gslab[j] += red[j];
  }
}

// ---------------------------------------------------------------------------
// vjp_kernel: vector-Jacobian product of ONE flow pass with respect to its
// input points -- the building block of the reference's autodiff helpers
// forward_jac / inverse_jac / gauge_potential (flows.py:203-211):
//   xbar[b,:] = ybar[b,:] . dF/dx(b)  +  ldbar[b] * d logdet/dx(b)
// Same forward-with-stash + backward as grad_kernel, without weight gradients.
// ---------------------------------------------------------------------------
struct VjpArgs {
  ModelArgs m;
  const float* pts;      // [B, D]
  const float* c;
  const float* ybar;     // [B, D] or null (= 0)
  const float* ldbar;    // [B] or null (= 0)
  float* xbar;           // [B, D] or null
  float* slabs;          // WGRAD: per-wave gradient slabs
  int64_t n_params;
  int64_t B, c_block;
  int32_t to_base;
  uint32_t div_magic;
  // finite-difference mode (cnf_logprob_fd_vjp): pts holds B / fd2 base points r_i; the pass differentiated is
  // log_prob at the evaluation point j = i * fd2 + 2 d + s (r_i -+ fd_h e_d) with seed +-gbar[i, d] * fd_inv_dx;
  // xbar[i, :] receives the SUM of the adjoints of r_i's fd2 evaluation points.  fd2 = 0: off.
  int32_t fd2;
  float fd_h, fd_inv_dx;
  const float* gbar;     // [B / fd2, D]
  // fused score term (cnf_score_fd_vjp): r1 != null.  pts = r3 (the samples at t), r1 / r2 the samples at t -+ dt/2; the
  // kernel forms the score from ITS forward passes, the residual u = (r2 - r1)/dt + coef score - drift(r3), the per-slice
  // sums of u^2 (sums), the adjoints of r1 / r2 (rbar1 / rbar2; xbar = the adjoint of r3) and seeds its own backward:
  // no gbar, and no separate forward launch over the 2 D evaluation points
  const float* r1; const float* r2;
  float* rbar1; float* rbar2;
  double* sums;
  float inv_dt, coef, drift_a, loss_coef;
  int32_t drift;         // CnfDrift or -1
};

// WGRAD=true additionally accumulates the parameter gradient of the pass (the
// backward of a differentiable flow op: cnf_pass_vjp).
template <bool FAST, bool WGRAD, int DFIX = 0>
__global__ __launch_bounds__(GTS_MAX, 2) void vjp_kernel(const VjpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int HDR = hdr_floats(GK);
  const int D = DFIX ? DFIX : a.m.D, L = a.m.L, DT = D * GROW;
  float* St = lds + PRE_FLOATS;
  float* Aa = St + (L + 1) * DT;
  float* Ab = Aa + DT;
  float* stage = Ab + DT + (threadIdx.x >> 6) * STAGE_FLOATS;       // (WGRAD = false: nothing is staged)
  float* gslab = WGRAD ? a.slabs + (int64_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (GTS >> 6) + (threadIdx.x >> 6))) * a.n_params : nullptr;
  if (WGRAD) slab_clear(gslab, a.n_params, threadIdx.x & 63);
  for (int i = threadIdx.x; i < HDR; i += GTS) lds[i] = a.m.prep[i];
  tile_consts(lds);
  const int tid = threadIdx.x;
  FirstAcc fa;      // WGRAD=false: written, never read: removed by the compiler
#pragma unroll
  for (int j = 0; j < GK; ++j) { fa.Wb[j] = 0.0f; fa.Hb[j] = 0.0f; }
#pragma unroll
  for (int j = 0; j <= GK; ++j) fa.Db[j] = 0.0f;
  // finite-difference mode: a tile holds whole groups of fd2 evaluation points (tp <= GTS of them)
  const int tp = a.fd2 ? (GTS / a.fd2) * a.fd2 : GTS;
  const int64_t n_tiles = (a.B + tp - 1) / tp;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t tile_start = tile * tp;
    const int64_t i = tile_start + tid;
    __syncthreads();
    float c, ld_bar;
    if (a.fd2) {
      const bool valid = tid < tp && i < a.B;
      const int64_t ib = valid ? i / a.fd2 : 0;
      const int k = (int)(i - ib * a.fd2), dd = k >> 1;
      for (int e = 0; e < D; ++e) {
        float v = valid ? a.pts[ib * D + e] : 0.0f;
        if (e == dd) v += (k & 1) ? -a.fd_h : a.fd_h;
        St[e * GROW + tid] = v;
        Aa[e * GROW + tid] = 0.0f;
      }
      const int64_t n_base = a.B / a.fd2;
      c = valid ? a.c[a.c_block >= n_base ? 0 : ib / a.c_block] : 0.0f;
      ld_bar = (valid && a.gbar) ? ((k & 1) ? -a.gbar[ib * D + dd] : a.gbar[ib * D + dd]) * a.fd_inv_dx : 0.0f;
    } else {
      tile_load1(a.pts, St, D, a.div_magic, tile_start, a.B);
      if (a.ybar) tile_load1(a.ybar, Aa, D, a.div_magic, tile_start, a.B);
      else for (int e = tid; e < DT; e += GTS) Aa[e] = 0.0f;
      c = i < a.B ? a.c[a.c_block >= a.B ? 0 : i / a.c_block] : 0.0f;
      ld_bar = (a.ldbar && i < a.B) ? a.ldbar[i] : 0.0f;
    }
    __syncthreads();
    const float ldsum = pass_fwd_stash<FAST, DFIX>(a.m, lds, St, c, a.to_base != 0);
    float ub_drift = 0.0f;      // fused score term: a * u_bar of this lane's (point, dimension) -- the OU drift's share of r3_bar
    if (a.fd2) {       // log_prob = sum -x^2/2 + ildj: the adjoint of the recovered base point is -ld_bar x
      const float* sL = St + L * DT + tid;
      if (a.r1) {      // the residual of this lane's (point, dimension) from the pair's two log_prob values
        const bool valid = tid < tp && i < a.B;
        const int64_t ib = valid ? i / a.fd2 : 0;
        const int k = (int)(i - ib * a.fd2), dd = k >> 1;
        const float lp = base_lp(sL, D) + ldsum;
        const float other = __shfl_xor(lp, 1, 64);             // (groups start at even lanes: the partner is lane ^ 1)
        const float score = ((k & 1) ? other - lp : lp - other) * a.fd_inv_dx;
        const int64_t o = ib * D + dd;
        const float r3d = valid ? a.pts[o] : 0.0f;
        const float vel = valid ? (a.r2[o] - a.r1[o]) * a.inv_dt : 0.0f;
        const float drift = a.drift == CNF_DRIFT_OU ? -a.drift_a * r3d : 0.0f;
        const float u = valid ? fmaf(score, a.coef, vel) - drift : 0.0f;
        const float ub = 2.0f * a.loss_coef * u;
        ld_bar = ((k & 1) ? -a.coef : a.coef) * ub * a.fd_inv_dx;
        ub_drift = a.drift == CNF_DRIFT_OU ? a.drift_a * ub : 0.0f;
        const bool owner = valid && !(k & 1);                  // one lane of the pair writes / counts
        if (owner) { a.rbar1[o] = -ub * a.inv_dt; a.rbar2[o] = ub * a.inv_dt; }
        // per-slice sums of u^2: lanes are ordered by point, so a wave's slices are those of its first and last valid lane
        const int64_t n_base = a.B / a.fd2;
        const long long slice = !valid ? -1 : (a.c_block >= n_base ? 0 : (long long)(ib / a.c_block));
        const int nv = (int)(a.B - tile_start < tp ? a.B - tile_start : tp) - (tid & ~63);      // valid lanes of this wave
        if (nv > 0) {
          const long long s_lo = __shfl(slice, 0, 64), s_hi = __shfl(slice, nv < 64 ? nv - 1 : 63, 64);
          float part = owner ? u * u : 0.0f;
          if (s_lo == s_hi) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
            if ((tid & 63) == 0) unsafeAtomicAdd(a.sums + s_lo, (double)part);
          } else if (owner) {
            unsafeAtomicAdd(a.sums + slice, (double)part);
          }
        }
      }
      for (int e = 0; e < D; ++e) Aa[e * GROW + tid] = -ld_bar * sL[e * GROW];
    }
    float* ain = pass_bwd<FAST, WGRAD, DFIX>(a.m, lds, St, Aa, Ab, ld_bar, c, a.to_base != 0, gslab, stage, fa);
    float* spare = ain == Aa ? Ab : Aa;            // the adjoint buffer the pass finished NOT in: scratch now
    if (a.fd2 && a.r1) spare[tid] = ub_drift;
    __syncthreads();
    if (a.fd2) {       // xbar[i, e] = sum over the fd2 evaluation points of base point i (fixed order: deterministic)
      if (a.xbar) {
        const int groups = tp / a.fd2;
        const int64_t first_base = tile_start / a.fd2, n_base = a.B / a.fd2;
        for (int idx = tid; idx < groups * D; idx += GTS) {
          const int il = idx / D, e = idx - il * D;
          if (first_base + il < n_base) {
            float sum = a.r1 ? spare[il * a.fd2 + 2 * e] : 0.0f;      // (+ the OU drift's -J^T u_bar = a u_bar, diagonal)
            for (int k = 0; k < a.fd2; ++k) sum += ain[e * GROW + il * a.fd2 + k];
            a.xbar[(first_base + il) * D + e] = sum;
          }
        }
      }
    } else if (a.xbar) {      // coalesced store of the input adjoints
      const int64_t base = tile_start * D;
      const int n_el = (int)(a.B - tile_start < GTS ? a.B - tile_start : GTS) * D;
      for (int e = tid; e < GTS * D; e += GTS) {
        const int s = a.div_magic ? (int)__umulhi((uint32_t)e, a.div_magic) : e, d = e - s * D;
        if (e < n_el) a.xbar[base + e] = ain[d * GROW + s];
      }
    }
  }
  if constexpr (WGRAD) {
    float red[GP];
#pragma unroll
    for (int j = 0; j < GK; ++j) { red[j] = fa.Wb[j]; red[GK + j] = fa.Hb[j]; }
#pragma unroll
    for (int j = 0; j <= GK; ++j) red[2 * GK + j] = fa.Db[j];
#pragma unroll
    for (int j = 0; j < GP; ++j) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) red[j] += __shfl_xor(red[j], off, 64);
    }
    if ((tid & 63) == 0) {
#pragma unroll
      for (int j = 0; j < GP; ++j) gslab[j] += red[j];
    }
  }
}

// ---------------------------------------------------------------------------
// The backward of ONE flow pass at dim 2 on the conditioner tables (cnf_pwl.h), with PER-PIECE SUFFICIENT
// STATISTICS in place of per-sample weight gradients.  On a piece of a slice's table both ReLU activity patterns
// are constant and theta, h2, h1 are affine in the conditioner's scalar input u, so every weight gradient is linear
// in  A = sum theta_bar  and  B = sum (u - u_ref) theta_bar  over the samples that land in the piece
// (oracle/pwl_grad.py restates the algebra; checked against per-sample backprop to 1e-9).  Per sample that leaves:
// the forward through the tables, the spline partials (cond_spline_bwd / table_spline_bwd as in the MLP kernels),
// the input adjoint S . theta_bar from the row already in LDS, and 20 accumulations -- no MLP recompute and no
// per-sample GEMM.  pwl_stats_finish_kernel turns the statistics into gradient slabs, once per piece.
//   vjp_pwl_kernel: two samples per lane (packed arithmetic), 512 threads; LDS = `first` table | L tables (PWL_LROWS-row window) |
//   64-bit fixed-point accumulators [L][64 pieces][33] (A | B per piece).
// ---------------------------------------------------------------------------
constexpr int PWL_STAT = 2 * PWL_P;                 // statistics per piece: A[16] | B[16]
// The statistics are accumulated in 64-bit FIXED POINT: measured on MI355X, ds_add_f32 retires a lane every ~2.6
// cycles whatever the addresses (the accumulation was 0.9 of 1.2 ms per 4.2 M-point pass, and neither bank padding
// nor replicated accumulators changed it), integer LDS atomics are ~15 x faster -- and integer sums do not depend
// on the order of the additions, so the gradient is bitwise reproducible like the slab scheme of the MLP kernels.
// Scale: 2^s with s = 28 - exponent of the largest |ybar|, |ldbar| of the call (adjoint_max_kernel): resolution
// 2^-28 of that magnitude (float32 itself resolves 2^-24), a single term may be 2^23 x it (the 1.5 2^52 rounding trick
// holds below 2^51), a sum 2^35 x it.
// In LDS a piece's 32 accumulators are 33 entries apart (entry m of piece p in bank pair (p + m) mod 32: lanes of
// different pieces do not collide); only the first PWL_ACC_W pieces have LDS accumulators (the rest -- far pieces,
// few samples -- go to global memory).
typedef unsigned long long stat_t;
constexpr int PWL_STAT_LDS = PWL_STAT + 1;
constexpr int PWL_ACC_W = 64;
constexpr int PWL_STAT_SLICES = 128;                // slices per chunk of the table backward (statistics buffer: 2 x 9.5 MB per flow layer)
// Lanes of a wave that land in the same piece add to the same addresses, and the LDS serialises them: the accumulators
// exist up to PWL_ACC_R times (as many as fit in LDS next to the L tables: 3 at L = 2, 1 from L = 3 on), lane t uses
// copy t mod acc_r (summed when a slice's statistics are flushed).  Measured at L = 2, one pass over 4.2 M points:
// 1 copy 0.34-0.39 ms, 3 copies 0.28-0.33 (scripts/exp_vjp_tables_only.py).
constexpr int PWL_ACC_R = 3;
// ... and what remains must not become a bank conflict instead: entry i of a piece sits in the 128-byte line slot
// (p + i) mod 16, so copy r starts PWL_ACC_SKEW r entries late and the B entries are stored rotated by 8 -- the six
// (copy, A | B) combinations of one piece and entry are six different slots.
constexpr int PWL_ACC_SKEW = 5;
__host__ __device__ constexpr int pwl_acc_entry(int e) { return e < PWL_P ? e : PWL_P + ((e - PWL_P + 8) & 15); }

// 2^(28 - e) for the largest adjoint magnitude 2^e <= |x| < 2^(e+1) (bits of |x| as given by adjoint_max_kernel);
// all-zero adjoints: any scale will do
__device__ __forceinline__ double stat_scale(uint32_t amax_bits) {
  int e = (int)(amax_bits >> 23) - 127;
  if (amax_bits == 0) e = 0;
  if (e < -100) e = -100;
  return __longlong_as_double((long long)(1023 + 28 - e) << 52);
}

// the largest |ybar|, |ldbar| of a call (bit pattern of a non-negative float: ordered like the integers)
__global__ __launch_bounds__(256) void adjoint_max_kernel(const float* __restrict__ ybar, int64_t n_y,
                                                         const float* __restrict__ ldbar, int64_t n_l, uint32_t* out) {
  uint32_t m = 0, bad = 0;      // bad: an Inf / NaN adjoint was seen -> out[1] (the table backward then answers NaN, like the MLP backward)
  auto take = [&](float v) { const uint32_t b = __float_as_uint(v) & 0x7fffffffu; bad |= b >= 0x7f800000u ? 1u : 0u; m = (b > m && b < 0x7f800000u) ? b : m; };
  auto scan = [&](const float* __restrict__ p, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {            // 16-byte loads, four in flight per thread
      const f4* q = reinterpret_cast<const f4*>(p);
      const int64_t n4 = n >> 2;
      int64_t i = t;
      for (; i + 3 * stride < n4; i += 4 * stride) {             // (explicitly: the compiler kept one load per iteration)
        const f4 v0 = q[i], v1 = q[i + stride], v2 = q[i + 2 * stride], v3 = q[i + 3 * stride];
#pragma unroll
        for (int e = 0; e < 4; ++e) { take(v0[e]); take(v1[e]); take(v2[e]); take(v3[e]); }
      }
      for (; i < n4; i += stride) { const f4 v = q[i]; take(v[0]); take(v[1]); take(v[2]); take(v[3]); }
      for (int64_t i = (n4 << 2) + t; i < n; i += stride) take(p[i]);
    } else {
      for (int64_t i = t; i < n; i += stride) take(p[i]);
    }
  };
  if (ybar) scan(ybar, n_y);
  if (ldbar) scan(ldbar, n_l);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
  __shared__ uint32_t wm[4];                 // one atomic per workgroup: they all land on one address
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t a01 = wm[0] > wm[1] ? wm[0] : wm[1], a23 = wm[2] > wm[3] ? wm[2] : wm[3];
    const uint32_t mm = a01 > a23 ? a01 : a23;
    if (mm) atomicMax(out, mm);
  }
  if (__builtin_amdgcn_ballot_w64(bad != 0) != 0 && (threadIdx.x & 63) == 0) atomicOr(out + 1, 1u);
}

struct VjpPwlArgs {
  ModelArgs m;
  const float* pts;      // [B, 2] inputs of the pass
  const float* ybar;     // [B, 2] or null
  const float* ldbar;    // [B] or null
  float* xbar;           // [B, 2] or null
  const float* tables;   // [n_slices][L][PWL_TBL]
  stat_t* stats;         // [n_slices][L][PWL_NPIECE][PWL_STAT] fixed point, zero on entry
  stat_t* coarse;        // the same shape at 2^-32 of the scale: terms too large for `stats` (ill-conditioned flows)
  uint32_t* amax;        // [0] bits of the largest |adjoint| of the call (adjoint_max_kernel), [1] non-finite flag
  float* slabs;          // slab b (this workgroup's): zeroed here, its first GP entries receive the `first` spline's
  int64_t n_params;      // per-bin adjoint sums (one owner per slab: a fixed summation order)
  int64_t B, slice_len;
  int32_t n_slices, tiles_per_slice;
  int32_t acc_r;         // copies of the LDS accumulators (1 .. PWL_ACC_R)
  // SEED (cnf_neg_logprob_vjp): the output adjoints are formed in the kernel -- ybar = seed_coef * base point,
  // ldbar = -seed_coef -- and sums[slice] receives sum -log_prob; amax_bits != 0: the scale's magnitude (no
  // adjoint_max_kernel ran: the seeds are bounded by what the kernel itself produces)
  float seed_coef;
  double* sums;
  uint32_t amax_bits;
  int32_t pts_shared;    // the slices share ONE set of points pts[slice_len, 2] (cnf_kinetic_potential_vjp)
};

constexpr int VJP_PWL_THREADS = 512;      // two samples per lane: tiles of 1 024 samples

// LFIX > 0: the number of flow layers is this constant (the reference's 2): the layer loops are unrolled, the kept
// inputs and outputs of the layers are registers with fixed names instead of arrays indexed by a loop counter
// (which the compiler served from select chains and 16 bytes of scratch)
template <bool TO_BASE, int LFIX = 0, bool SEED = false>
__global__ __launch_bounds__(VJP_PWL_THREADS) void vjp_pwl_kernel(const VjpPwlArgs a) {
  static_assert(!SEED || TO_BASE, "the density-fit term: data -> base");
  constexpr int K = GK, WIN = PWL_LROWS;
  constexpr bool INV = !TO_BASE;
  constexpr int MAXL = LFIX ? LFIX : 4;
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  constexpr int HDR = (hdr_floats(K) + 3) & ~3;
  const int L = LFIX ? LFIX : a.m.L, NT = blockDim.x, tid = threadIdx.x, TS = 2 * NT;
  float* tab = lds_raw;
  float* tbl = lds_raw + HDR;
  stat_t* acc = reinterpret_cast<stat_t*>(tbl + L * pwl_ltbl(WIN));      // [L][PWL_ACC_W][PWL_STAT_LDS] (8-byte aligned: HDR, pwl_ltbl even)
  const int acc_n = L * PWL_ACC_W * PWL_STAT_LDS;                          // one copy
  const int acc_r = a.acc_r;
  float* red = reinterpret_cast<float*>(acc + acc_r * (acc_n + PWL_ACC_SKEW));      // [waves][GP]
  for (int i = tid; i < hdr_floats(K); i += NT) tab[i] = a.m.prep[i];
  for (int i = tid; i < acc_r * (acc_n + PWL_ACC_SKEW); i += NT) acc[i] = 0;
  stat_t* acc_mine = acc + (tid % acc_r) * (acc_n + PWL_ACC_SKEW);
  const SplineConsts sc = sc_scalars(a.m.sc);
  // fixed-point scale 2^(28 - e), e = the exponent of the largest adjoint; x -> round(x scale) by the 1.5 2^52 trick
  const double fx_scale = SliceSum::uniform(stat_scale(a.amax_bits ? a.amax_bits : *a.amax));      // (wave-uniform: a scalar register pair)
  [[maybe_unused]] SliceSum ssum;
  auto to_fixed = [&](double xs) -> stat_t {              // xs = x scale, |xs| < 2^50
    const double d = xs + 6755399441055744.0;
    return (stat_t)(__double_as_longlong(d) - __double_as_longlong(6755399441055744.0));
  };
  // one term into accumulator `e` of (slice, layer l, piece p): LDS for the first PWL_ACC_W pieces; a term beyond the
  // fine scale's reach (|x| >= 2^22 x the largest adjoint: only ill-conditioned flows produce them) goes, 2^32
  // coarser, to the second statistics array in global memory
  auto accumulate = [&](int slice, int l, int p, int e, float x) {
    const double xs = (double)x * fx_scale;
    if (fabs(xs) < 1125899906842624.0) {                     // 2^50
      const stat_t q = to_fixed(xs);
      if (p < PWL_ACC_W) {
        typedef stat_t __attribute__((address_space(3))) * lds_q_ptr;
        lds_q_ptr d3 = (lds_q_ptr)(uintptr_t)(uint32_t)(uintptr_t)(acc_mine + (l * PWL_ACC_W + p) * PWL_STAT_LDS + pwl_acc_entry(e));
        __hip_atomic_fetch_add(d3, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        atomicAdd(a.stats + (((int64_t)slice * L + l) * PWL_NPIECE + p) * PWL_STAT + e, q);
      }
    } else if (fabs(xs) < 4835703278458516698824704.0) {     // 2^82
      atomicAdd(a.coarse + (((int64_t)slice * L + l) * PWL_NPIECE + p) * PWL_STAT + e, to_fixed(xs * 2.3283064365386963e-10));
    }                                                         // (beyond: not representable in float32 sums either)
  };
  // One sample's theta_bar into the statistics of its piece.  The adjoints of a softmax group's logits sum to zero: the
  // last width and the last height entry are not accumulated (pwl_stats_finish_kernel restores them as minus the sum
  // of the other four) -- 20 atomic instructions per sample and layer.  One range test decides between the
  // straight-line form (every term within the fine scale, the piece's accumulators in LDS) and the general one.
  auto add_stats = [&](int slice, int l, int p, float du, const float (&t)[2 * K], int kk, float sb0, float sb1) {
    float big = fmaxf(fabsf(sb0), fabsf(sb1));
#pragma unroll
    for (int m2 = 0; m2 < 2 * K; ++m2) big = fmaxf(big, fabsf(t[m2]));
    big *= fmaxf(1.0f, fabsf(du));
    // (fmaxf drops NaNs: the sum of the terms does not) a non-finite term poisons the call's gradient, like the
    // float accumulation of the MLP backward would -- the fixed-point sums cannot carry it themselves
    float chk = (sb0 + sb1) * du;
#pragma unroll
    for (int m2 = 0; m2 < 2 * K; ++m2) chk += t[m2];
    if (!(fabsf(chk) < INFINITY) || !(big < INFINITY)) {
      atomicOr(a.amax + 1, 1u);
    } else if (p < PWL_ACC_W && (double)big * fx_scale < 1125899906842624.0) {
      typedef stat_t __attribute__((address_space(3))) * lds_q_ptr;
      lds_q_ptr d3 = (lds_q_ptr)(uintptr_t)(uint32_t)(uintptr_t)(acc_mine + (l * PWL_ACC_W + p) * PWL_STAT_LDS);
      auto add = [&](lds_q_ptr d, float x) {
        const double dd = fma((double)x, fx_scale, 6755399441055744.0);
        __hip_atomic_fetch_add(d, (stat_t)(__double_as_longlong(dd) - __double_as_longlong(6755399441055744.0)),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      };
      // odd lanes add their B entry where even lanes add their A entry: lanes of one piece meet on two addresses per
      // instruction instead of one
      const bool sw = tid & 1;
      const float f0 = sw ? du : 1.0f, f1 = sw ? 1.0f : du;
      // (entry e of A at e, of B at pwl_acc_entry(16 + e) = 16 + (e + 8) mod 16: for the lane, B - A is +24 below e = 8, +8 from there)
      static_assert(2 * K + K <= PWL_P && pwl_acc_entry(PWL_P + 8) == PWL_P, "slope entries 10 .. 15 do not wrap");
      const lds_q_ptr qa = d3 + (sw ? PWL_P + 8 : 0), qb = d3 + (sw ? 0 : PWL_P + 8);      // entries 0 .. 7
      const lds_q_ptr ra = d3 + (sw ? PWL_P - 8 : 0), rb = d3 + (sw ? 0 : PWL_P - 8);      // entries 8 .. 15
      // ... and lanes 2, 3 (mod 4) take the entries of each pair (0,1) (2,3) (5,6) (7,8) (slopes) in the other order: twelve
      // slots per piece and instruction pair.  A selected value, and the lane's address moved by one entry.
      const bool sx = tid & 2;
      const int up = sx ? 1 : 0;
      auto add_pair = [&](lds_q_ptr a0, lds_q_ptr b0, int i, float x, float y) {      // x -> entry i, y -> entry i + 1
        const float u0 = sx ? y : x, u1 = sx ? x : y;
        add(a0 + i + up, f0 * u0);     add(b0 + i + up, f1 * u0);
        add(a0 + i + 1 - up, f0 * u1); add(b0 + i + 1 - up, f1 * u1);
      };
      add_pair(qa, qb, 0, t[0], t[1]); add_pair(qa, qb, 2, t[2], t[3]);
      add_pair(qa, qb, 5, t[5], t[6]);
      // (entries 7 | 8 straddle the rotation of the B half)
      {
        const float u0 = sx ? t[8] : t[7], u1 = sx ? t[7] : t[8];
        add((sx ? ra : qa) + 7 + up, f0 * u0);     add((sx ? rb : qb) + 7 + up, f1 * u0);
        add((sx ? qa : ra) + 8 - up, f0 * u1);     add((sx ? qb : rb) + 8 - up, f1 * u1);
      }
      add_pair(ra, rb, 2 * K + kk, sb0, sb1);
    } else {
#pragma unroll
      for (int m2 = 0; m2 < 2 * K; ++m2)
        if (m2 % K != K - 1) { accumulate(slice, l, p, m2, t[m2]); accumulate(slice, l, p, PWL_P + m2, du * t[m2]); }
      accumulate(slice, l, p, 2 * K + kk, sb0);     accumulate(slice, l, p, PWL_P + 2 * K + kk, du * sb0);
      accumulate(slice, l, p, 2 * K + kk + 1, sb1); accumulate(slice, l, p, PWL_P + 2 * K + kk + 1, du * sb1);
    }
  };
  // the `first` spline's per-bin adjoint sums of the lane's samples (pair-wide; folded at the end)
  v2f Wb[K], Hb[K], Db[K + 1];
#pragma unroll
  for (int j = 0; j < K; ++j) { Wb[j] = splat<v2f>(0.0f); Hb[j] = splat<v2f>(0.0f); }
#pragma unroll
  for (int j = 0; j <= K; ++j) Db[j] = splat<v2f>(0.0f);

  auto flush = [&](int slice) {          // LDS accumulators -> the slice's statistics, and clear
    __syncthreads();
    stat_t* g = a.stats + (int64_t)slice * L * PWL_NPIECE * PWL_STAT;
    for (int i = tid; i < L * PWL_ACC_W * PWL_STAT; i += NT) {
      const int lp = i / PWL_STAT, m2 = i - lp * PWL_STAT;            // (layer, piece), entry
      const int l = lp / PWL_ACC_W, p = lp - l * PWL_ACC_W;
      stat_t* e = acc + lp * PWL_STAT_LDS + pwl_acc_entry(m2);
      stat_t v = 0;
      for (int r = 0; r < acc_r; ++r) { v += e[r * (acc_n + PWL_ACC_SKEW)]; e[r * (acc_n + PWL_ACC_SKEW)] = 0; }
      if (v != 0) atomicAdd(g + ((int64_t)l * PWL_NPIECE + p) * PWL_STAT + m2, v);
    }
    __syncthreads();
  };

  const int total = a.n_slices * a.tiles_per_slice;
  const int per_block = (total + gridDim.x - 1) / gridDim.x;
  const int t0 = blockIdx.x * per_block;
  const int t1 = t0 + per_block < total ? t0 + per_block : total;
  int cur = -1;
  // A tile's points and adjoints are requested one tile ahead: issued where they are used, the loads' ~2 us were a
  // third of a wave's time (rocprofv3 counters of round 3: 37 % of the wave cycles in s_waitcnt, 3 % of them on LDS)
  struct TileIn { f4 x, yb; v2f ldb; };
  auto tile_geom = [&](int tile, int& slice, int64_t& g, bool& v0, bool& v1) {
    slice = tile / a.tiles_per_slice;
    const int64_t s0 = (int64_t)slice * a.slice_len;
    const int64_t len = a.B - s0 < a.slice_len ? a.B - s0 : a.slice_len;
    const int64_t j = (int64_t)(tile - slice * a.tiles_per_slice) * TS + 2 * tid;      // the lane's samples: j, j + 1
    g = s0 + j;
    v0 = j < len; v1 = j + 1 < len;
  };
  auto tile_load = [&](int tile) {
    TileIn t;
    t.x = f4{0.f, 0.f, 0.f, 0.f}; t.yb = t.x; t.ldb = v2f{0.f, 0.f};
    int slice; int64_t g; bool v0, v1;
    tile_geom(tile, slice, g, v0, v1);
    const int64_t gp = a.pts_shared ? g - (int64_t)slice * a.slice_len : g;
    if (v1) {
      t.x = *reinterpret_cast<const f4*>(a.pts + 2 * gp);
      if (a.ybar) t.yb = *reinterpret_cast<const f4*>(a.ybar + 2 * g);
      if (a.ldbar) t.ldb = *reinterpret_cast<const v2f*>(a.ldbar + g);
    } else if (v0) {
      t.x[0] = a.pts[2 * gp]; t.x[1] = a.pts[2 * gp + 1];
      if (a.ybar) { t.yb[0] = a.ybar[2 * g]; t.yb[1] = a.ybar[2 * g + 1]; }
      if (a.ldbar) t.ldb.x = a.ldbar[g];
    }
    return t;
  };
  TileIn nxt;
  if (t0 < t1) nxt = tile_load(t0);
  for (int tile = t0; tile < t1; ++tile) {
    int slice; int64_t g; bool v0, v1;
    tile_geom(tile, slice, g, v0, v1);
    if (slice != cur) {
      if (cur >= 0) flush(cur); else __syncthreads();
      pwl_stage<WIN>(tbl, a.tables + (int64_t)slice * L * PWL_TBL, L, tid, NT);
      cur = slice;
      __syncthreads();
    }
    const f4 x = nxt.x, yb = nxt.yb;
    v2f ld_bar = nxt.ldb;
    if (tile + 1 < t1) nxt = tile_load(tile + 1);
    __builtin_amdgcn_sched_barrier(0);
    // a non-finite POINT would pass unnoticed (the table search and the splines' clamps turn it into some finite
    // point): it poisons the call's gradient like a non-finite adjoint does
    if (!(fabsf((x[0] + x[1]) + (x[2] + x[3])) < INFINITY)) atomicOr(a.amax + 1, 1u);
    v2f u0 = {x[0], x[2]}, u1 = {x[1], x[3]};
    v2f ob0 = {yb[0], yb[2]}, ob1 = {yb[1], yb[3]};
    [[maybe_unused]] v2f lacc = splat<v2f>(0.0f);      // SEED: the pass's log|det J|
    const float* gtbl = a.tables + (int64_t)slice * L * PWL_TBL;
    // ---- forward through the tables, keeping every layer's inputs and outputs (flow2_tables' calls)
    v2f in_f[MAXL], in_o[MAXL], out_f[MAXL];
#pragma unroll
    for (int step = 0; step < MAXL; ++step) {
      if (step < L) {
        const int l = TO_BASE ? L - 1 - step : step;
        const bool odd = l & 1;
        const v2f uf = odd ? u1 : u0, uo = odd ? u0 : u1;
        v2f of, oo = splat<v2f>(0.0f), ld;
        table_spline<K, INV, true, v2f>(tab, uf, sc, of, ld);
        if constexpr (SEED) lacc += ld;
        // The backward never reads a layer's conditioned OUTPUT: data -> base takes the spline partials at the layer's
        // input, base -> data at the output, which cond_spline_bwd_rows forms itself in the bin it selects.  So the last
        // layer's table lookup, softmax and spline -- whose result nothing downstream reads -- are skipped here
        if (SEED || step != L - 1) {      // (SEED: the term's value and seeds need the whole pass)
          const float* tl = tbl + l * pwl_ltbl(WIN);
          const float* gl = gtbl + (int64_t)l * PWL_TBL;
          const v2f uc = TO_BASE ? of : uf;
          PwlRows rr;
          bool general;
          v2f qa[K], qb[K];
          pwl_find<WIN>(tl, uc, rr, general);
          rr.dua = pwl_logit_pairs<WIN>(rr.ra, gl, rr.pa, uc.x, qa);
          rr.dub = pwl_logit_pairs<WIN>(rr.rb, gl, rr.pb, uc.y, qb);
          auto slopes = [&](int ka, int kb, v2f& ta, v2f& tb) {
            ta = pwl_slope_pair<WIN>(rr.ra, gl, rr.pa, ka, rr.dua);
            tb = pwl_slope_pair<WIN>(rr.rb, gl, rr.pb, kb, rr.dub);
          };
          cond_spline_rows<K, INV, true, false, false>(qa, qb, slopes, uo, sc, oo, ld);
          if constexpr (SEED) lacc += ld;
        }
        in_f[step] = uf; in_o[step] = uo; out_f[step] = of;
        u0 = odd ? oo : of; u1 = odd ? of : oo;
      }
    }
    if constexpr (SEED) {
      // -log_prob = |x|^2 / 2 + log 2 pi - ildj at the recovered base point x; d(seed_coef * sum) / d(x, ildj)
      const v2f nlp = vfma(u0, u0, u1 * u1) * 0.5f + (float)(2 * HALF_LOG_2PI) - lacc;
      float part = (v0 ? nlp.x : 0.0f) + (v1 ? nlp.y : 0.0f);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
      ssum.add(a.sums, slice, part);
      ob0 = v2f{v0 ? u0.x : 0.0f, v1 ? u0.y : 0.0f} * a.seed_coef;
      ob1 = v2f{v0 ? u1.x : 0.0f, v1 ? u1.y : 0.0f} * a.seed_coef;
      ld_bar = v2f{v0 ? -a.seed_coef : 0.0f, v1 ? -a.seed_coef : 0.0f};
    }
    // ---- backward
#pragma unroll
    for (int step = MAXL - 1; step >= 0; --step) {
      if (step < L) {
        const int l = TO_BASE ? L - 1 - step : step;
        const bool odd = l & 1;
        v2f ob_f = odd ? ob1 : ob0;
        const v2f ob_o = odd ? ob0 : ob1;
        const float* tl = tbl + l * pwl_ltbl(WIN);
        const float* gl = gtbl + (int64_t)l * PWL_TBL;
        const v2f uc = TO_BASE ? out_f[step] : in_f[step];
        PwlRows rr;
        bool general;
        v2f qa[K], qb[K];
        pwl_find<WIN>(tl, uc, rr, general);
        rr.dua = pwl_logit_pairs<WIN>(rr.ra, gl, rr.pa, uc.x, qa);
        rr.dub = pwl_logit_pairs<WIN>(rr.rb, gl, rr.pb, uc.y, qb);
        auto slopes = [&](int ka, int kb, v2f& ta, v2f& tb) {
          ta = pwl_slope_pair<WIN>(rr.ra, gl, rr.pa, ka, rr.dua);
          tb = pwl_slope_pair<WIN>(rr.rb, gl, rr.pb, kb, rr.dub);
        };
        // theta_bar: the 2K softmax entries in tb, the two non-zero slope entries (knots kk, kk + 1 of the selected
        // bin) apart -- their accumulators and their rows' slopes are addressed with kk
        v2f tb[2 * K], kkf, sb0, sb1;
        const v2f ub_o = cond_spline_bwd_rows<K, INV, true>(qa, qb, slopes, in_o[step], ob_o, ld_bar, sc, tb, kkf, sb0, sb1,
                                                            __builtin_amdgcn_ballot_w64(general) == 0);
        const int kka = (int)kkf.x, kkb = (int)kkf.y;
        float ta[2 * K], tc[2 * K];
#pragma unroll
        for (int m2 = 0; m2 < 2 * K; ++m2) { ta[m2] = tb[m2].x; tc[m2] = tb[m2].y; }
        const v2f ucond_bar = v2f{pwl_row_dot<WIN>(rr.ra, gl, rr.pa, kka, ta, sb0.x, sb1.x),
                                  pwl_row_dot<WIN>(rr.rb, gl, rr.pb, kkb, tc, sb0.y, sb1.y)};
        v2f ub_f = splat<v2f>(0.0f);
        if (TO_BASE) ob_f += ucond_bar; else ub_f = ucond_bar;
        ub_f += table_spline_bwd<K, INV, true>(tab, in_f[step], ob_f, ld_bar, sc, Wb, Hb, Db);
        // (last: LDS operations complete in order, a read issued behind the 40 accumulations would wait for all of them)
        if (v0) add_stats(slice, l, rr.pa, rr.dua, ta, kka, sb0.x, sb1.x);
        if (v1) add_stats(slice, l, rr.pb, rr.dub, tc, kkb, sb0.y, sb1.y);
        ob0 = odd ? ub_o : ub_f; ob1 = odd ? ub_f : ub_o;
      }
    }
    if (a.xbar) {
      if (v1) *reinterpret_cast<f4*>(a.xbar + 2 * g) = f4{ob0.x, ob1.x, ob0.y, ob1.y};
      else if (v0) { a.xbar[2 * g] = ob0.x; a.xbar[2 * g + 1] = ob1.x; }
    }
  }
  if (cur >= 0) flush(cur);
  if constexpr (SEED) ssum.flush();
  // the `first` spline's per-bin adjoint sums: wave shuffle, block reduce, one slab entry per block
  {
    float r[GP];
#pragma unroll
    for (int j = 0; j < GK; ++j) { r[j] = Wb[j].x + Wb[j].y; r[GK + j] = Hb[j].x + Hb[j].y; }
#pragma unroll
    for (int j = 0; j <= GK; ++j) r[2 * GK + j] = Db[j].x + Db[j].y;
#pragma unroll
    for (int j = 0; j < GP; ++j) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) r[j] += __shfl_xor(r[j], off, 64);
    }
    if ((tid & 63) == 0) {
#pragma unroll
      for (int j = 0; j < GP; ++j) red[(tid >> 6) * GP + j] = r[j];
    }
    __syncthreads();
    float* slab = a.slabs + (int64_t)blockIdx.x * a.n_params;
    for (int64_t i = GP + tid; i < a.n_params; i += NT) slab[i] = 0.0f;
    if (tid < GP) {
      float s = 0.0f;
      for (int w2 = 0; w2 < (NT >> 6); ++w2) s += red[w2 * GP + tid];
      slab[tid] = s;
    }
  }
}

// One block per (slice, layer): the slice's per-piece statistics -> that layer's 592 conditioner gradients, written
// (with zeros elsewhere) into slab first_slab + block; the statistics are cleared for the next call.
struct StatsFinishArgs {
  const float* weights;  // prep + hdr: the conditioner weights snapshot
  int64_t per_layer;
  const float* cvals;    // [n_slices]
  const float* tables;
  stat_t* stats;
  stat_t* coarse;
  const uint32_t* amax;  // [0] largest adjoint, [1] non-finite flag
  uint32_t amax_bits;    // != 0: in place of amax[0] (VjpPwlArgs)
  float* slabs;          // [first_slab + n_slices * L][n_params]
  int64_t n_params;
  int32_t L, first_slab;
  int32_t split;         // blocks per (slice, layer): block g of them takes the pieces of every split-th group of four
};

__global__ __launch_bounds__(256) void pwl_stats_finish_kernel(const StatsFinishArgs a) {
  constexpr int H = PWL_H, P = PWL_P, NW = 2 * H + H + H * H + H + H * P + P;      // 592 floats of one conditioner
  constexpr int NWV = 4;                                   // waves: one piece per wave at a time
  constexpr int PER_LANE = (NW + 63) / 64;                 // 10 outputs per lane
  __shared__ float av[H], bv[H], W1[H * H], b1[H], Wo[H * P];
  __shared__ float stat[PWL_NPIECE * PWL_STAT];            // the slice-layer's statistics, read once (37 KB)
  __shared__ float scr[NWV][9 * H];                        // per wave: Bu | m1a | m1b | Pk | Qk | G2 | G2u | G1 | G1u
  __shared__ float part[NWV][NW];                          // the waves' partial results
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // (few slices -- the density-fit term has one -- would leave the chip to ns x L workgroups walking 289 pieces each)
  const int sl = blockIdx.x / a.split, gsub = blockIdx.x - sl * a.split;
  const int slice = sl / a.L, l = sl % a.L;
  const float* w = a.weights + l * a.per_layer;
  const float c = a.cvals[slice];
  if (tid < H) {
    av[tid] = w[H + tid];
    bv[tid] = fmaf(w[tid], c, w[2 * H + tid]);
    b1[tid] = w[3 * H + H * H + tid];
  }
  for (int i = tid; i < H * H; i += blockDim.x) { W1[i] = w[3 * H + i]; Wo[i] = w[3 * H + H * H + H + i]; }
  const float* T = a.tables + (int64_t)sl * PWL_TBL;
  stat_t* st = a.stats + (int64_t)sl * PWL_NPIECE * PWL_STAT;
  const int n = __float_as_int(T[PWL_N_SLOT]);
  const double inv_scale = 1.0 / stat_scale(a.amax_bits ? a.amax_bits : *a.amax);
  stat_t* sc2 = a.coarse + (int64_t)sl * PWL_NPIECE * PWL_STAT;
  auto mine = [&](int p) { return (p / NWV) % a.split == gsub; };
  for (int i = tid; i < (n + 1) * PWL_STAT; i += blockDim.x) {
    if (!mine(i / PWL_STAT)) continue;
    const stat_t q = st[i], qc = sc2[i];
    stat[i] = (float)(((double)(long long)q + (double)(long long)qc * 4294967296.0) * inv_scale);
    if (q != 0) st[i] = 0;                                 // cleared for the next call
    if (qc != 0) sc2[i] = 0;
  }
  __syncthreads();
  // vjp_pwl_kernel leaves out the last logit of each softmax group (widths, heights): the group's adjoints sum to zero
  for (int i = tid; i < (n + 1) * 4; i += blockDim.x) {
    if (!mine(i >> 2)) continue;
    float* g = stat + (i >> 2) * PWL_STAT + (i & 2 ? P : 0) + (i & 1 ? GK : 0);
    g[GK - 1] = -((g[0] + g[1]) + (g[2] + g[3]));
  }
  __syncthreads();
  float out[PER_LANE];
#pragma unroll
  for (int q = 0; q < PER_LANE; ++q) out[q] = 0.0f;
  float* Bu = scr[wv], *m1a = Bu + H, *m1b = m1a + H, *Pk = m1b + H, *Qk = Pk + H, *G2 = Qk + H, *G2u = G2 + H,
        *G1 = G2u + H, *G1u = G1 + H;
  for (int p = wv + NWV * gsub; p <= n; p += NWV * a.split) {      // wave-uniform loop: no block barrier inside
    const float* A = stat + p * PWL_STAT;
    const float sv = lane < PWL_STAT ? A[lane] : 0.0f;
    if (__builtin_amdgcn_ballot_w64(sv != 0.0f) == 0) continue;
    const float lo = p == 0 ? -INFINITY : T[p - 1], hi = p < n ? T[p] : INFINITY;
    const bool fl = lo > -INFINITY, fh = hi < INFINITY;
    const float ut = fl && fh ? 0.5f * (lo + hi) : (fl ? lo + 1.0f : (fh ? hi - 1.0f : 0.0f));
    const float uref = T[PWL_OFF_REF + p];
    float on1 = 0.0f;
    if (lane < H) {
      Bu[lane] = fmaf(uref, A[lane], A[P + lane]);          // sum (u - u_ref) g  ->  sum u g
      on1 = fmaf(av[lane], ut, bv[lane]) > 0.0f ? 1.0f : 0.0f;
      m1a[lane] = on1 * av[lane]; m1b[lane] = on1 * bv[lane];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    if (lane < H) {
      const int k = lane;
      float Ps = 0.0f, Qs = b1[k], ha = 0.0f, hb = 0.0f;
      for (int j = 0; j < H; ++j) { Ps = fmaf(W1[j * H + k], m1a[j], Ps); Qs = fmaf(W1[j * H + k], m1b[j], Qs); }
      for (int m = 0; m < P; ++m) { ha = fmaf(Wo[k * P + m], A[m], ha); hb = fmaf(Wo[k * P + m], Bu[m], hb); }
      const float on2 = fmaf(Ps, ut, Qs) > 0.0f ? 1.0f : 0.0f;
      Pk[k] = on2 * Ps; Qk[k] = on2 * Qs; G2[k] = on2 * ha; G2u[k] = on2 * hb;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    if (lane < H) {
      float ga = 0.0f, gb = 0.0f;
      for (int k = 0; k < H; ++k) { ga = fmaf(W1[lane * H + k], G2[k], ga); gb = fmaf(W1[lane * H + k], G2u[k], gb); }
      G1[lane] = on1 * ga; G1u[lane] = on1 * gb;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < PER_LANE; ++q) {
      const int o = lane + q * 64;
      if (o < NW) {
        float v;
        if (o < H) v = c * G1[o];                                            // W0[c row]
        else if (o < 2 * H) v = G1u[o - H];                                  // W0[u row]
        else if (o < 3 * H) v = G1[o - 2 * H];                               // b0
        else if (o < 3 * H + H * H) { const int e = o - 3 * H, j = e / H, k = e % H; v = fmaf(m1a[j], G2u[k], m1b[j] * G2[k]); }
        else if (o < 4 * H + H * H) v = G2[o - 3 * H - H * H];               // b1
        else if (o < 4 * H + H * H + H * P) { const int e = o - 4 * H - H * H, k = e / P, m = e % P; v = fmaf(Pk[k], Bu[m], Qk[k] * A[m]); }
        else v = A[o - 4 * H - H * H - H * P];                               // bo
        out[q] += v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();      // scr is reused by the next piece
  }
#pragma unroll
  for (int q = 0; q < PER_LANE; ++q) { const int o = lane + q * 64; if (o < NW) part[wv][o] = out[q]; }
  __syncthreads();
  // slab 1 + block: zeros except this layer's conditioner
  float* slab = a.slabs + (int64_t)(a.first_slab + blockIdx.x) * a.n_params;
  const int64_t base = GP + (int64_t)l * a.per_layer;
  for (int64_t i = tid; i < a.n_params; i += blockDim.x) {
    const int64_t o = i - base;
    slab[i] = (o >= 0 && o < NW) ? (part[0][o] + part[1][o]) + (part[2][o] + part[3][o]) : 0.0f;
  }
  if (a.amax[1] != 0) {      // a non-finite adjoint or term somewhere in the call: the gradient says so
    for (int64_t i = tid; i < a.n_params; i += blockDim.x) slab[i] = __int_as_float(0x7fc00000);
  }
}

// grad[p] += sum over slabs; the first 16 entries are per-bin adjoint sums of
// the `first` spline and go through the softmax / softplus Jacobians (float64).
// Block = 32 parameters x 32 slab stripes (a thread that walks all slabs alone
// pays one dependent L2 round trip per slab: 1-2 ms for 2 048 slabs).
__global__ __launch_bounds__(1024) void grad_finish_kernel(const float* __restrict__ slabs, int64_t n_slabs,
                                                           int64_t n_params, const float* __restrict__ params,
                                                           float* __restrict__ grad, double span_eff,
                                                           double sp_offset, uint32_t* clear2) {
  // clear2: two words this launch leaves zero for the next call (the table backward's adjoint maximum and non-finite
  // flag: every reader ran before this kernel on the same stream)
  if (clear2 && blockIdx.x == 0 && threadIdx.x < 2) clear2[threadIdx.x] = 0;
  __shared__ float part[32][33];
  __shared__ double raw[GP];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t p = (int64_t)blockIdx.x * 32 + tx;
  float acc = 0.0f;
  if (p < n_params) {
    int64_t b = ty;
    for (; b + 96 < n_slabs; b += 128) {       // four independent loads in flight
      const float v0 = slabs[b * n_params + p], v1 = slabs[(b + 32) * n_params + p];
      const float v2 = slabs[(b + 64) * n_params + p], v3 = slabs[(b + 96) * n_params + p];
      acc += (v0 + v1) + (v2 + v3);
    }
    for (; b < n_slabs; b += 32) acc += slabs[b * n_params + p];
  }
  part[ty][tx] = acc;
  __syncthreads();
  if (ty == 0) {
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 32; ++k) s += part[k][tx];
    if (blockIdx.x == 0 && tx < GP) raw[tx] = (double)s;
    else if (p < n_params) grad[p] += s;
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x < GP) {
    const int j = threadIdx.x;
    double g;
    if (j < 2 * GK) {
      const int pt = j / GK, jj = j % GK;
      double mx = params[pt * GK];
      for (int k = 1; k < GK; ++k) mx = fmax(mx, (double)params[pt * GK + k]);
      double pr[GK], sum = 0.0, dot = 0.0;
      for (int k = 0; k < GK; ++k) { pr[k] = exp((double)params[pt * GK + k] - mx); sum += pr[k]; }
      for (int k = 0; k < GK; ++k) { pr[k] /= sum; dot += raw[pt * GK + k] * pr[k]; }
      g = span_eff * pr[jj] * (raw[j] - dot);
    } else {
      g = raw[j] / (1.0 + exp(-((double)params[j] + sp_offset)));
    }
    grad[j] += (float)g;
  }
}

// optax.adam(lr): mu = b1 mu + (1-b1) g; nu = b2 nu + (1-b2) g^2;
// update = -lr * (mu / (1-b1^t)) / (sqrt(nu / (1-b2^t)) + eps)   (eps_root = 0)
__global__ void adam_kernel(float* __restrict__ params, const float* __restrict__ grad, float* __restrict__ mu,
                            float* __restrict__ nu, int64_t n, float lr, float b1, float b2, float eps,
                            float bc1, float bc2) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = grad[i];
  const float m = b1 * mu[i] + (1.0f - b1) * g;
  const float v = b2 * nu[i] + (1.0f - b2) * g * g;
  mu[i] = m; nu[i] = v;
  params[i] -= lr * (m / bc1) / (sqrtf(v / bc2) + eps);
}

// The same update with the step count read from device memory (state[0] of cnf_step_begin, already incremented for
// this step): the bias corrections cannot be kernel arguments of a step that is captured once and replayed
__global__ void adam_dev_kernel(float* __restrict__ params, const float* __restrict__ grad, float* __restrict__ mu,
                                float* __restrict__ nu, int64_t n, float lr, float b1, float b2, float eps,
                                const uint64_t* __restrict__ state) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float step = (float)state[0];
  const float bc1 = 1.0f - powf(b1, step), bc2 = 1.0f - powf(b2, step);
  const float g = grad[i];
  const float m = b1 * mu[i] + (1.0f - b1) * g;
  const float v = b2 * nu[i] + (1.0f - b2) * g * g;
  mu[i] = m; nu[i] = v;
  params[i] -= lr * (m / bc1) / (sqrtf(v / bc2) + eps);
}

// out[0] = sum_i v[i] w[i]: the composite loss from the terms' per-slice sums and their coefficients, in a fixed order
__global__ __launch_bounds__(256) void weighted_sum_kernel(const double* __restrict__ v, const double* __restrict__ w,
                                                           int64_t n, double* __restrict__ out) {
  __shared__ double part[256];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) acc += v[i] * w[i];
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if ((int)threadIdx.x < s2) part[threadIdx.x] += part[threadIdx.x + s2];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = part[0];
}

// ---------------------------------------------------------------------------
// Epilogues of the UNFUSED loss terms (dim >= applications.UNFUSED_SCORE_MIN_DIM: the flow passes are separate,
// chip-filling launches; these turn their outputs into per-slice sums and, for value_and_grad, into the adjoints
// the backward launches start from).  One thread per sample.
// ---------------------------------------------------------------------------
struct ResidArgs {
  const float* r;        // [3 n, D]: r1 (t - dt/2) | r2 (t + dt/2) | r3 (t)
  const float* score;    // [n, D] central-difference score of log_prob at r3
  float* rbar;           // [3 n, D] or null
  float* sbar;           // [n, D] or null
  double* sums;          // [n / count]
  int64_t n, count;
  int32_t D, subtype;    // subtype < 0: no drift (kinetic_with_score), else CnfDrift
  float inv_dt, coef, a, loss_coef;
};

// sum_d ((r2 - r1)/dt + coef score_d - drift_d(r3))^2 per slice (applications.py:245-374) and its adjoints
__global__ __launch_bounds__(256) void score_residual_kernel(const ResidArgs a) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int D = a.D;
  float acc = 0.0f;
  const bool valid = i < a.n;
  if (valid) {
    const float* r1 = a.r + i * D;
    const float* r2 = a.r + (a.n + i) * D;
    const float* r3 = a.r + (2 * a.n + i) * D;
    const float* sc = a.score + i * D;
    float x = r3[0], y = D > 1 ? r3[1] : 0.0f, z = D > 2 ? r3[2] : 0.0f;
    float ub0 = 0.f, ub1 = 0.f, ub2 = 0.f;
    for (int d = 0; d < D; ++d) {
      float drift = 0.0f;
      switch (a.subtype) {
        case CNF_DRIFT_OU: drift = -a.a * r3[d]; break;
        case CNF_DRIFT_SMILE: { const float q = x * x + y * y - 4.0f; drift = (d == 0 ? -q * x : -q * y - (y - 1.0f) * 2.0f) * a.a; break; }
        case CNF_DRIFT_NONGRADIENT: drift = d == 0 ? x * -a.a - y * 0.5f : y * -a.a + x * 0.5f; break;
        case CNF_DRIFT_LORENZ: drift = d == 0 ? (y - x) * 10.0f : (d == 1 ? x * 9.0f * (28.0f / 9.0f - z) - y : x * 9.0f * y - z * (8.0f / 3.0f)); break;
        default: break;
      }
      const float u = fmaf(sc[d], a.coef, (r2[d] - r1[d]) * a.inv_dt) - drift;
      acc = fmaf(u, u, acc);
      if (a.rbar) {
        const float ub = 2.0f * a.loss_coef * u;
        a.rbar[i * D + d] = -ub * a.inv_dt;
        a.rbar[(a.n + i) * D + d] = ub * a.inv_dt;
        a.sbar[i * D + d] = a.coef * ub;
        if (d == 0) ub0 = ub; else if (d == 1) ub1 = ub; else if (d == 2) ub2 = ub;
        // r3_bar = -J_drift^T u_bar; the diagonal OU field is complete here, the coupled 2-D / 3-D fields below
        a.rbar[(2 * a.n + i) * D + d] = a.subtype == CNF_DRIFT_OU ? a.a * ub : 0.0f;
      }
    }
    if (a.rbar) {
      float* r3b = a.rbar + (2 * a.n + i) * D;
      if (a.subtype == CNF_DRIFT_SMILE) {
        const float q = x * x + y * y - 4.0f;
        r3b[0] = -(ub0 * (-a.a * (q + 2.0f * x * x)) + ub1 * (-a.a * 2.0f * x * y));
        r3b[1] = -(ub0 * (-a.a * 2.0f * x * y) + ub1 * (-a.a * (q + 2.0f * y * y + 2.0f)));
      } else if (a.subtype == CNF_DRIFT_NONGRADIENT) {
        r3b[0] = -(ub0 * (-a.a) + ub1 * 0.5f);
        r3b[1] = -(ub0 * (-0.5f) + ub1 * (-a.a));
      } else if (a.subtype == CNF_DRIFT_LORENZ) {
        r3b[0] = -(ub0 * -10.0f + ub1 * (28.0f - 9.0f * z) + ub2 * 9.0f * y);
        r3b[1] = -(ub0 * 10.0f - ub1 + ub2 * 9.0f * x);
        r3b[2] = -(ub1 * (-9.0f * x) + ub2 * (-8.0f / 3.0f));
      }
    }
  }
  // a wave whose samples share a slice: shuffle reduce, one double atomic; otherwise one atomic per sample
  const int64_t slice = valid ? i / a.count : -1;
  const int64_t s0 = __shfl(slice, 0, 64), s63 = __shfl(slice, 63, 64);
  if (s0 == s63 && s0 >= 0) {
    float part = acc;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(a.sums + s0, (double)part);
  } else if (valid) {
    atomicAdd(a.sums + slice, (double)acc);
  }
}

// Value and adjoints of the terms cnf_ot_amd.applications composes from table launches at dim 2 (5.4b): one thread
// per sample, per-slice sums like score_residual_kernel.
//   CNF_TERM_KINETIC:      r = [r1 | r2] (2 n points), sums[s] = sum |(r2 - r1) / dt|^2, rbar = -+ 2 c (r2 - r1) / dt^2
//   CNF_TERM_POTENTIAL:    r (n points), sums[s] = sum V(r), rbar = c grad V(r)            (CnfPotential subtype, a)
//   CNF_TERM_NEG_LOGPROB:  r = recovered base points x, aux = ildj: sums[s] = -sum (base(x) + ildj), rbar = c x,
//                          auxbar = -c
struct TermResidArgs {
  const float* r; const float* aux; float* rbar; float* auxbar; double* sums;
  int64_t n, count;
  int32_t kind, D, subtype;
  float p0, loss_coef;
  uint32_t* amax;      // non-null (DFIX = 2, rbar written): [0] max |rbar| as bits, [1] non-finite flag -- what adjoint_max_kernel
};                     // would find in rbar afterwards (cnf_kinetic_potential_vjp)

// DFIX = 2: the points are float2 -- one 8-byte load / store per point instead of a strided loop.
// The kernel is a stream of independent 8-byte accesses: 1 024-thread workgroups, RESID_PER points per thread a
// workgroup-width apart (one point per thread left the chip waiting on workgroup launches: 49 us for 4.2 M points,
// twice the time of the memory traffic).  The sums: a lane adds up its points of the wave's current slice; when the
// slice changes the wave flushes (shuffles + one double atomic); at the end lanes -> wave -> workgroup (LDS) -> ONE
// atomic per run of equal slices in the workgroup.  (One atomic per wave put 65 536 of them on the single address of
// a 4.2 M-sample slice and took longer than the flow kernels around it.)
constexpr int RESID_PER = 4;
template <int DFIX>
__global__ __launch_bounds__(1024) void term_residual_kernel(const TermResidArgs a) {
  const int D = DFIX ? DFIX : a.D;
  __shared__ double wsum[16];
  __shared__ long long wslice[16];
  uint32_t amx = 0, abad = 0;
  auto seen = [&](v2f gbar) {
    const uint32_t bx = __float_as_uint(gbar.x) & 0x7fffffffu, by = __float_as_uint(gbar.y) & 0x7fffffffu;
    const uint32_t b = bx > by ? bx : by;
    abad |= b >= 0x7f800000u ? 1u : 0u;
    amx = (b > amx && b < 0x7f800000u) ? b : amx;
  };
  auto term = [&](int64_t i) -> float {      // the term's value at point i; writes the adjoints
    float v = 0.0f;
    if (DFIX == 2) {
      const v2f* r2p = reinterpret_cast<const v2f*>(a.r);
      v2f* rb2 = reinterpret_cast<v2f*>(a.rbar);
      if (a.kind == CNF_TERM_KINETIC) {
        const float inv_dt = 1.0f / a.p0, g = 2.0f * a.loss_coef * inv_dt * inv_dt;
        const v2f dr = r2p[a.n + i] - r2p[i];
        const v2f w = dr * inv_dt;
        v = fmaf(w.x, w.x, w.y * w.y);
        if (a.rbar) { rb2[i] = dr * -g; rb2[a.n + i] = dr * g; if (a.amax) seen(dr * g); }
      } else if (a.kind == CNF_TERM_POTENTIAL) {
        const v2f x = r2p[i];
        const float s2 = fmaf(x.x, x.x, x.y * x.y);
        const v2f xm = x - a.p0, xp = x + a.p0;
        const float sm = fmaf(xm.x, xm.x, xm.y * xm.y), sp = fmaf(xp.x, xp.x, xp.y * xp.y);
        v2f gr;
        if (a.subtype == CNF_POT_DOUBLE_WELL) { v = 0.25f * sm * sp; gr = (xm * sp + xp * sm) * 0.5f; }      // applications.py:184-188
        else if (a.subtype == CNF_POT_OBSTACLE) { v = 50.0f * expf(-0.5f * s2); gr = x * -v; }                // :190-191
        else { v = 0.5f * s2; gr = x; }                                                                        // :181-182
        if (a.rbar) { rb2[i] = gr * a.loss_coef; if (a.amax) seen(gr * a.loss_coef); }
      } else {      // CNF_TERM_NEG_LOGPROB
        const v2f x = r2p[i];
        if (a.rbar) rb2[i] = x * a.loss_coef;
        v = -(a.aux[i] - 0.5f * fmaf(x.x, x.x, x.y * x.y) - (float)(2 * HALF_LOG_2PI));
        if (a.auxbar) a.auxbar[i] = -a.loss_coef;
      }
    } else {
      if (a.kind == CNF_TERM_KINETIC) {
        const float inv_dt = 1.0f / a.p0, g = 2.0f * a.loss_coef * inv_dt * inv_dt;
        for (int d = 0; d < D; ++d) {
          const float dr = a.r[(a.n + i) * D + d] - a.r[i * D + d];
          const float w = dr * inv_dt;
          v = fmaf(w, w, v);
          if (a.rbar) { a.rbar[i * D + d] = -g * dr; a.rbar[(a.n + i) * D + d] = g * dr; }
        }
      } else if (a.kind == CNF_TERM_POTENTIAL) {
        float s2 = 0.0f, sm = 0.0f, sp = 0.0f;
        for (int d = 0; d < D; ++d) {
          const float x = a.r[i * D + d];
          s2 = fmaf(x, x, s2); sm = fmaf(x - a.p0, x - a.p0, sm); sp = fmaf(x + a.p0, x + a.p0, sp);
        }
        if (a.subtype == CNF_POT_DOUBLE_WELL) v = 0.25f * sm * sp;
        else if (a.subtype == CNF_POT_OBSTACLE) v = 50.0f * expf(-0.5f * s2);
        else v = 0.5f * s2;
        if (a.rbar) {
          for (int d = 0; d < D; ++d) {
            const float x = a.r[i * D + d];
            float gr;
            if (a.subtype == CNF_POT_DOUBLE_WELL) gr = 0.5f * ((x - a.p0) * sp + (x + a.p0) * sm);
            else if (a.subtype == CNF_POT_OBSTACLE) gr = -v * x;
            else gr = x;
            a.rbar[i * D + d] = a.loss_coef * gr;
          }
        }
      } else {
        float s2 = 0.0f;
        for (int d = 0; d < D; ++d) {
          const float x = a.r[i * D + d];
          s2 = fmaf(x, x, s2);
          if (a.rbar) a.rbar[i * D + d] = a.loss_coef * x;
        }
        v = -(a.aux[i] - 0.5f * s2 - (float)(D * HALF_LOG_2PI));
        if (a.auxbar) a.auxbar[i] = -a.loss_coef;
      }
    }
    return v;
  };
  auto wave_sum = [](float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
  };
  const bool small = a.n < ((int64_t)1 << 31) && a.count < ((int64_t)1 << 31);      // 32-bit slice arithmetic
  const int64_t base = blockIdx.x * (int64_t)(blockDim.x * RESID_PER) + threadIdx.x;
  float acc = 0.0f;              // this lane's points of slice `cur`
  long long cur = -1;            // wave-uniform
#pragma unroll
  for (int it = 0; it < RESID_PER; ++it) {
    const int64_t i = base + (int64_t)it * blockDim.x;
    const bool valid = i < a.n;
    const float v = valid ? term(i) : 0.0f;
    const long long slice = !valid ? -1 : small ? (long long)((uint32_t)i / (uint32_t)a.count) : (long long)(i / a.count);
    const long long sl0 = __shfl(slice, 0, 64), sl63 = __shfl(slice, 63, 64);
    if (sl0 == sl63) {                      // (-1 == -1: a wave past the end, contributing nothing)
      if (sl0 != cur) {
        const float part = wave_sum(acc);
        if (cur >= 0 && (threadIdx.x & 63) == 0) unsafeAtomicAdd(a.sums + cur, (double)part);
        cur = sl0; acc = 0.0f;
      }
      acc += v;
    } else if (valid) {
      unsafeAtomicAdd(a.sums + slice, (double)v);      // a wave across a slice border: per lane
    }
  }
  __shared__ uint32_t wmax[16];
  if (DFIX == 2 && a.amax) {      // adjoint_max_kernel's result, without its pass over rbar
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(amx, off, 64); amx = o > amx ? o : amx; }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amx;
    if (__builtin_amdgcn_ballot_w64(abad != 0) != 0 && (threadIdx.x & 63) == 0) atomicOr(a.amax + 1, 1u);
  }
  const float part = wave_sum(acc);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { wsum[wv] = (double)part; wslice[wv] = cur; }
  __syncthreads();
  if (DFIX == 2 && a.amax && threadIdx.x == 64) {
    // one atomic per workgroup, and none where the maximum on record is already as large (a wave's atomic each put
    // 130 000 of them on one address: +0.25 ms for config 5's kinetic + obstacle term)
    uint32_t mm = 0;
    for (int w2 = 0; w2 < (int)(blockDim.x >> 6); ++w2) mm = wmax[w2] > mm ? wmax[w2] : mm;
    if (mm > __hip_atomic_load(a.amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.amax, mm);
  }
  if (threadIdx.x == 0) {
    const int nw = blockDim.x >> 6;
    double run = 0.0;
    long long c2 = -1;
    for (int w2 = 0; w2 < nw; ++w2) {
      if (wslice[w2] != c2) { if (c2 >= 0) unsafeAtomicAdd(a.sums + c2, run); c2 = wslice[w2]; run = 0.0; }
      run += wsum[w2];
    }
    if (c2 >= 0) unsafeAtomicAdd(a.sums + c2, run);
  }
}

struct RklArgs {
  const float* y;        // [n, D] samples
  const float* lp;       // [n] their log_prob
  float* ybar;           // [n, D] or null
  float* lpbar;          // [n] or null
  double* sums;          // [1]
  int64_t n;
  int32_t D;
  float t, T, beta, loss_coef;
};

// sum_i log_prob_i - log(N(y_i; 0, vs I) ws + N(y_i; 0, vt I) wt)  (reverse_kl_loss_fn, applications.py:129-163)
__global__ __launch_bounds__(256) void rkl_residual_kernel(const RklArgs a) {
  // grid-stride with a running per-lane sum: one double atomic per wave at the END (one per 64 samples put 65 536
  // atomics on the single address of a 4 M-sample batch)
  float acc = 0.0f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
    const int D = a.D;
    float s2 = 0.0f;
    for (int d = 0; d < D; ++d) { const float r = a.y[i * D + d]; s2 = fmaf(r, r, s2); }
    const float vs = 2.0f / a.beta * (a.T + 1.0f), vt = 2.0f / a.beta;
    const float ws = (a.T - a.t) / a.T, wt = a.t / a.T;
    const float ls = -0.5f * D * logf(6.283185307179586f * vs), lt = -0.5f * D * logf(6.283185307179586f * vt);
    const float as = -0.5f * s2 / vs + ls, at = -0.5f * s2 / vt + lt;
    const float mx = fmaxf(as, at);
    const float es = expf(as - mx) * ws, et = expf(at - mx) * wt;
    acc += a.lp[i] - (mx + logf(es + et));
    if (a.ybar) {
      const float g = (es / vs + et / vt) / (es + et);      // -d logmix / d y_e = g * y_e
      for (int d = 0; d < D; ++d) a.ybar[i * D + d] = a.loss_coef * g * a.y[i * D + d];
      a.lpbar[i] = a.loss_coef;
    }
  }
  float part = acc;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
  if ((threadIdx.x & 63) == 0) unsafeAtomicAdd(a.sums, (double)part);
}

}  // namespace cnf

using namespace cnf;

#undef GTS
static size_t grad_lds_bytes(int D, int L, int ts = GTS_MAX, bool generic = true) {
  // table, ones / zeros, condition column + noise + (L+1) stashes + 2 adjoint buffers + velocity, r3, r3_bar, u_bar
  // (rows of ts + 4 floats) + one MFMA staging area per wave
  return (size_t)(hdr_floats(GK) + 64 + (ts + 4) + D * (ts + 4) * (1 + (L + 1) + 6) + (ts / 64) * STAGE_FLOATS +
                  (generic ? GP * (ts + 4) : 0)) * sizeof(float);     // (generic-dimension kernel: + rows for the `first` spline's accumulators)
}
static size_t vjp_lds_bytes(int D, int L, int ts, bool wgrad) {
  // (without weight gradients nothing is staged)
  return (size_t)(hdr_floats(GK) + 64 + (ts + 4) + D * (ts + 4) * ((L + 1) + 2) + (wgrad ? (ts / 64) * STAGE_FLOATS : 0)) * sizeof(float);
}
// The tile size (threads per workgroup) that puts the most waves on a CU: LDS allows 160 KB / lds(ts) workgroups
// of ts / 64 waves; the backward kernels' ~230 VGPRs allow 8 waves (2 per SIMD).  Ties go to the larger tile.
template <class F> static int pick_tile(F lds_of) {
  int best = 64, best_waves = 0;
  for (int ts = GTS_MAX; ts >= 64; ts >>= 1) {
    const size_t b = lds_of(ts);
    if (b > 160 * 1024) continue;
    int waves = (int)((160 * 1024) / b) * (ts / 64);
    if (waves > 8) waves = 8;
    if (waves > best_waves) { best_waves = waves; best = ts; }
  }
  return best;
}

extern "C" int cnf_grad_supported(const CnfConfig* c) {
  // hidden 16 / 2 hidden layers / 5 bins (the MFMA weight-gradient tiles are 16x16), dim <= 14 (the
  // first layer's inputs + bias row fit 16 MFMA rows), and the tile's LDS working set within one CU
  return c && !c->periodized && c->hidden_size == 16 && c->mlp_num_layers == 2 && c->num_bins == 5 && c->dim >= 1 && c->dim <= 14 &&
         c->num_layers >= 1 && grad_lds_bytes(c->dim, c->num_layers, 64) <= 160 * 1024;
}

static int check_term(const CnfModel* m, const CnfLossSpec* spec) {
  if (!spec || spec->kind < CNF_TERM_KINETIC || spec->kind > CNF_TERM_NEG_LOGPROB) return CNF_ERR_INVALID;
  const int D = m->cfg.dim;
  if (spec->kind <= CNF_TERM_FLOW_MATCHING && !(spec->dt > 0.f)) return CNF_ERR_INVALID;
  if ((spec->kind == CNF_TERM_KINETIC_SCORE || spec->kind == CNF_TERM_FLOW_MATCHING) && !(spec->dx > 0.f)) return CNF_ERR_INVALID;
  if (spec->kind == CNF_TERM_FLOW_MATCHING) {
    if ((spec->subtype == CNF_DRIFT_SMILE || spec->subtype == CNF_DRIFT_NONGRADIENT) && D != 2) return CNF_ERR_INVALID;
    if (spec->subtype == CNF_DRIFT_LORENZ && D != 3) return CNF_ERR_INVALID;
  }
  return CNF_OK;
}

extern "C" int cnf_loss_terms_grad_multi(CnfModel* m, int32_t n_terms, const CnfLossSpec* specs, const float* const* pts,
                                         const int32_t* pts_shared, const float* const* t, const int64_t* n_slices,
                                         const int64_t* B, const float* scale, double* const* sums, float* grad,
                                         const float* params, void* stream_) {
  if (!m || n_terms < 1 || n_terms > GRAD_MAX_JOBS || !specs || !pts || !pts_shared || !t || !n_slices || !B || !scale ||
      !sums || !grad || !params)
    return CNF_ERR_INVALID;
  if (!m->params_set) return CNF_ERR_INVALID;
  if (!cnf_grad_supported(&m->cfg)) return CNF_ERR_UNSUPPORTED;
  const int D = m->cfg.dim, L = m->cfg.num_layers;
  hipStream_t stream = (hipStream_t)stream_;
  const bool generic = !(m->fast_math && D == 2);          // (grad_kernel<true, 2> keeps everything in registers)
  const int ts = pick_tile([&](int tt) { return grad_lds_bytes(D, L, tt, generic); });
  const size_t lds = grad_lds_bytes(D, L, ts, generic);
  if (lds > 160 * 1024) return CNF_ERR_UNSUPPORTED;
  GradArgs a;
  a.m = model_args(m); a.slabs = m->grad_slabs; a.n_params = m->n_params; a.div_magic = m->div_magic;
  a.n_jobs = 0; a.n_tiles = 0;
  for (int i = 0; i < n_terms; ++i) {
    const int r = check_term(m, specs + i);
    if (r != CNF_OK) return r;
    if (!pts[i] || !t[i] || !sums[i] || n_slices[i] < 0 || B[i] < 0) return CNF_ERR_INVALID;
    if (n_slices[i] == 0) continue;
    if (hipMemsetAsync(sums[i], 0, sizeof(double) * (size_t)n_slices[i], stream) != hipSuccess) return CNF_ERR_HIP;
    if (B[i] == 0) continue;
    GradJob& J = a.job[a.n_jobs++];
    J.spec = specs[i]; J.pts = pts[i]; J.t = t[i]; J.sums = sums[i];
    J.B = B[i]; J.n_slices = n_slices[i]; J.pts_slice_stride = pts_shared[i] ? 0 : B[i];
    J.scale = scale[i]; J.first_tile = a.n_tiles;
    a.n_tiles += ((B[i] + ts - 1) / ts) * n_slices[i];
  }
  if (a.n_jobs == 0) return CNF_OK;
  for (int i = a.n_jobs; i < GRAD_MAX_JOBS; ++i) { a.job[i] = a.job[0]; a.job[i].first_tile = a.n_tiles; }
  if (!m->grad_slabs) return CNF_ERR_INVALID;      // cnf_grad_enable first
  if (wait_for_params(m, stream) != CNF_OK) return CNF_ERR_HIP;
  const int64_t max_grid = m->grad_max_blocks * 4 / (ts / 64);           // (slabs: one per wave)
  int64_t n_slabs = 0;
  auto launch = [&](auto kernel) -> bool {
    if (!ensure_lds(kernel, lds)) return false;
    int64_t cap = (int64_t)resident_blocks_per_cu(kernel, ts, lds) * m->num_cus;
    if (cap > max_grid) cap = max_grid;
    const int64_t grid = balanced_grid(a.n_tiles, cap);
    n_slabs = grid * (ts / 64);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(ts), lds, stream, a);
    return true;
  };
  if (m->fast_math) {
    if (D == 2 ? !launch(grad_kernel<true, 2>) : !launch(grad_kernel<true>)) return CNF_ERR_HIP;
  } else {
    if (!launch(grad_kernel<false>)) return CNF_ERR_HIP;
  }
  if (hipGetLastError() != hipSuccess) return CNF_ERR_HIP;
  const int fb = (int)((m->n_params + 31) / 32);
  hipLaunchKernelGGL(grad_finish_kernel, dim3(fb), dim3(1024), 0, stream, m->grad_slabs, n_slabs, m->n_params, params,
                     grad, (double)m->sc.span_eff, (double)m->sc.sp_offset, (uint32_t*)nullptr);
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

extern "C" int cnf_loss_terms_grad(CnfModel* m, const CnfLossSpec* spec, const float* pts, int pts_shared,
                                   const float* t, int64_t n_slices, int64_t B, float scale, double* sums,
                                   float* grad, const float* params, void* stream_) {
  if (!spec || !pts || !t || !sums) return CNF_ERR_INVALID;
  const int32_t shared = pts_shared ? 1 : 0;
  return cnf_loss_terms_grad_multi(m, 1, spec, &pts, &shared, &t, &n_slices, &B, &scale, &sums, grad, params, stream_);
}

extern "C" int cnf_grad_enable(CnfModel* m, int64_t max_blocks) {
  if (!m) return CNF_ERR_INVALID;
  if (!cnf_grad_supported(&m->cfg)) return CNF_ERR_UNSUPPORTED;
  if (max_blocks <= 0) max_blocks = (int64_t)m->num_cus * 2;
  if (m->grad_slabs && m->grad_max_blocks >= max_blocks) return CNF_OK;
  if (m->grad_slabs) { (void)hipFree(m->grad_slabs); m->grad_slabs = nullptr; }
  if (hipMalloc((void**)&m->grad_slabs, sizeof(float) * (size_t)(max_blocks * 4 * m->n_params)) != hipSuccess) return CNF_ERR_NOMEM;
  m->grad_max_blocks = max_blocks;
  // the table backward of dim-2 passes (vjp_pwl_kernel): per-piece statistics of PWL_STAT_SLICES slices, kept zero
  // between calls (pwl_stats_finish_kernel clears what it reads)
  if (!m->pwl_stats && m->cfg.dim == 2 && m->cfg.num_layers <= 4 && m->cfg.mlp_num_layers == 2) {
    // [64 bytes: the call's largest adjoint][statistics][coarse statistics]
    const size_t bytes = 64 + 2 * sizeof(stat_t) * (size_t)PWL_STAT_SLICES * m->cfg.num_layers * PWL_NPIECE * PWL_STAT;
    if (hipMalloc((void**)&m->pwl_stats, bytes) == hipSuccess) {
      if (hipMemset(m->pwl_stats, 0, bytes) != hipSuccess) { (void)hipFree(m->pwl_stats); m->pwl_stats = nullptr; }
    } else {
      m->pwl_stats = nullptr;          // (the MLP backward remains)
    }
  }
  return CNF_OK;
}

extern "C" int cnf_adam_step(float* params, const float* grad, float* mu, float* nu, int64_t n, float lr, float b1,
                             float b2, float eps, int64_t step, void* stream) {
  if (!params || !grad || !mu || !nu || n < 0 || step < 1) return CNF_ERR_INVALID;
  if (n == 0) return CNF_OK;
  const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grad,
                     mu, nu, n, lr, b1, b2, eps, bc1, bc2);
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

extern "C" int cnf_adam_step_dev(float* params, const float* grad, float* mu, float* nu, int64_t n, float lr, float b1,
                                 float b2, float eps, const uint64_t* state, void* stream) {
  if (!params || !grad || !mu || !nu || !state || n < 0) return CNF_ERR_INVALID;
  if (n == 0) return CNF_OK;
  hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grad,
                     mu, nu, n, lr, b1, b2, eps, state);
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

extern "C" int cnf_weighted_sum(const double* v, const double* w, int64_t n, double* out, void* stream) {
  if (!v || !w || !out || n < 0) return CNF_ERR_INVALID;
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, v, w, n, out);
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

// The table form of cnf_pass_vjp (vjp_pwl_kernel + pwl_stats_finish_kernel + grad_finish_kernel): dim 2, the
// reference's network, a condition uniform over slices, tables reserved on the stream (cnf_model_reserve) for at
// least min(n_slices, PWL_STAT_SLICES) slices.  CNF_ERR_UNSUPPORTED: the caller runs the MLP backward.
// seed_sums != null: the density-fit form (cnf_neg_logprob_vjp) -- no adjoints come in, the kernel seeds itself with
// seed_coef and writes the slices' sums of -log_prob.
// built != null (cnf_kinetic_potential_vjp): the slices' tables are built (one chunk), the adjoint maximum is in place
// (cnf_term_residual's kernel left it), and `pts` is ONE slice of points that every slice reads.
static int pass_vjp_pwl(CnfModel* m, int to_base, const float* pts, const float* c, int64_t c_block,
                        const float* ybar, const float* ldbar, float* xbar, float* grad, const float* params,
                        int64_t B, hipStream_t stream, float seed_coef = 0.0f, double* seed_sums = nullptr,
                        const float* built = nullptr) {
  const CnfConfig& g = m->cfg;
  if (!m->use_pwl || !m->fast_math || !m->pwl_stats || g.dim != 2 || g.hidden_size != PWL_H || g.num_bins != GK ||
      g.mlp_num_layers != 2 || g.num_layers > 4 || g.periodized)
    return CNF_ERR_UNSUPPORTED;
  const int L = g.num_layers;
  const int64_t slice_len = c_block < B ? c_block : B;
  const int64_t n_slices = (B + slice_len - 1) / slice_len;
  // worth it while a slice amortises its tables and their per-piece finishing (measured crossover: see DESIGN.md)
  // (measured, scripts/exp_vjp_crossover.py, round 3's kernel: 131 072 points 0.049 vs 0.044 ms for the MLP backward,
  // 262 144 points 0.056 vs 0.066 -- 32 slices of 8 192: 0.062 vs 0.066 --, 524 288 points 0.072 vs 0.110; slices of
  // 4 096 points lose until there are ~100 of them.  Round 2's kernel crossed over at twice that.)
  if (m->use_pwl == 1 && (slice_len < 8192 || B < 262144)) return CNF_ERR_UNSUPPORTED;
  // (a lane's two samples are one 16-byte access of the points and adjoints, one 8-byte access of ldbar)
  if ((reinterpret_cast<uintptr_t>(pts) & 15) || (reinterpret_cast<uintptr_t>(ybar) & 15) ||
      (reinterpret_cast<uintptr_t>(xbar) & 15) || (reinterpret_cast<uintptr_t>(ldbar) & 7) || (n_slices > 1 && (slice_len & 1)))
    return CNF_ERR_UNSUPPORTED;
  const int threads = VJP_PWL_THREADS, tile = 2 * threads;
  auto lds_bytes = [&](int acc_r) {
    return sizeof(float) * (size_t)(((hdr_floats(GK) + 3) & ~3) + L * pwl_ltbl(PWL_LROWS) +
                                    2 * acc_r * (L * PWL_ACC_W * PWL_STAT_LDS + PWL_ACC_SKEW) + (threads / 64) * GP);
  };
  int acc_r = PWL_ACC_R;
  while (acc_r > 1 && lds_bytes(acc_r) > 160 * 1024) --acc_r;
  const size_t lds = lds_bytes(acc_r);
  if (lds > 160 * 1024) return CNF_ERR_UNSUPPORTED;
  const bool seeded = seed_sums != nullptr;
  if (seeded ? (L == 2 ? !ensure_lds(vjp_pwl_kernel<true, 2, true>, lds) : !ensure_lds(vjp_pwl_kernel<true, 0, true>, lds))
      : L == 2 ? (to_base ? !ensure_lds(vjp_pwl_kernel<true, 2>, lds) : !ensure_lds(vjp_pwl_kernel<false, 2>, lds))
               : (to_base ? !ensure_lds(vjp_pwl_kernel<true>, lds) : !ensure_lds(vjp_pwl_kernel<false>, lds)))
    return CNF_ERR_UNSUPPORTED;
  // The seeds are seed_coef x (a base point | 1): the scale is set for adjoints up to 32 |seed_coef| -- base points
  // lie within the splines' range of +-10 unless the data does not -- with the 2^22 of head room every term has
  uint32_t amax_bits = 0;
  if (seeded) {
    const float bound = 32.0f * fabsf(seed_coef);
    union { float f; uint32_t u; } bits; bits.f = bound; amax_bits = bits.u;
    if (!(bound < INFINITY)) return CNF_ERR_INVALID;
    if (hipMemsetAsync(seed_sums, 0, sizeof(double) * (size_t)n_slices, stream) != hipSuccess) return CNF_ERR_HIP;
  }
  const int64_t tps = (slice_len + tile - 1) / tile;
  if (built && n_slices > PWL_STAT_SLICES) return CNF_ERR_UNSUPPORTED;
  for (int64_t s0 = 0; s0 < n_slices; s0 += PWL_STAT_SLICES) {
    const int64_t ns = n_slices - s0 < PWL_STAT_SLICES ? n_slices - s0 : PWL_STAT_SLICES;
    float* tables = const_cast<float*>(built);
    const int r = built ? CNF_OK : cnf_internal_build_tables(m, stream, c + s0, ns, &tables);
    if (r != CNF_OK) return s0 == 0 ? r : CNF_ERR_HIP;          // (a later chunk cannot fail on its own)
    m->last_path = CNF_PATH_TABLES;
    const int64_t first = s0 * slice_len;
    VjpPwlArgs a;
    a.m = model_args(m);
    a.pts = pts + (built ? 0 : 2 * first); a.pts_shared = built ? 1 : 0; a.ybar = ybar ? ybar + 2 * first : nullptr; a.ldbar = ldbar ? ldbar + first : nullptr;
    a.xbar = xbar ? xbar + 2 * first : nullptr;
    uint32_t* amax = reinterpret_cast<uint32_t*>(m->pwl_stats);
    stat_t* stats = reinterpret_cast<stat_t*>(reinterpret_cast<char*>(m->pwl_stats) + 64);
    stat_t* coarse = stats + (size_t)PWL_STAT_SLICES * L * PWL_NPIECE * PWL_STAT;
    a.tables = tables; a.stats = stats; a.coarse = coarse; a.amax = amax; a.slabs = m->grad_slabs; a.n_params = m->n_params;
    a.B = (B - first) < ns * slice_len ? (B - first) : ns * slice_len;
    a.slice_len = slice_len; a.n_slices = (int32_t)ns; a.tiles_per_slice = (int32_t)tps; a.acc_r = acc_r;
    a.seed_coef = seed_coef; a.sums = seeded ? seed_sums + s0 : nullptr; a.amax_bits = amax_bits;
    // (amax is zero on entry: cleared at allocation and by the grad_finish_kernel launch of the previous chunk or call)
    if (!seeded && !built)
      hipLaunchKernelGGL(adjoint_max_kernel, dim3((unsigned)(m->num_cus * 4)), dim3(256), 0, stream, a.ybar, a.ybar ? 2 * a.B : 0, a.ldbar,
                       a.ldbar ? a.B : 0, amax);
    const int64_t tiles = ns * tps;
    const int64_t grid = tiles < m->num_cus ? tiles : m->num_cus;
    const int split = ns * L <= 32 ? 8 : 1;
    if (grid + ns * L * split > m->grad_max_blocks * 4) return s0 == 0 ? CNF_ERR_UNSUPPORTED : CNF_ERR_HIP;      // (slabs: cnf_grad_enable)
    if (seeded) {
      if (L == 2) hipLaunchKernelGGL((vjp_pwl_kernel<true, 2, true>), dim3((unsigned)grid), dim3(threads), lds, stream, a);
      else hipLaunchKernelGGL((vjp_pwl_kernel<true, 0, true>), dim3((unsigned)grid), dim3(threads), lds, stream, a);
    } else if (L == 2) {
      if (to_base) hipLaunchKernelGGL((vjp_pwl_kernel<true, 2>), dim3((unsigned)grid), dim3(threads), lds, stream, a);
      else hipLaunchKernelGGL((vjp_pwl_kernel<false, 2>), dim3((unsigned)grid), dim3(threads), lds, stream, a);
    } else {
      if (to_base) hipLaunchKernelGGL(vjp_pwl_kernel<true>, dim3((unsigned)grid), dim3(threads), lds, stream, a);
      else hipLaunchKernelGGL(vjp_pwl_kernel<false>, dim3((unsigned)grid), dim3(threads), lds, stream, a);
    }
    StatsFinishArgs f;
    f.weights = m->prep + hdr_floats(GK); f.per_layer = m->per_layer; f.cvals = c + s0; f.tables = tables;
    f.stats = stats; f.coarse = coarse; f.amax = amax; f.amax_bits = amax_bits; f.slabs = m->grad_slabs; f.n_params = m->n_params; f.L = L;
    f.first_slab = (int32_t)grid; f.split = split;
    hipLaunchKernelGGL(pwl_stats_finish_kernel, dim3((unsigned)(ns * L * split)), dim3(256), 0, stream, f);
    const int fb = (int)((m->n_params + 31) / 32);
    hipLaunchKernelGGL(grad_finish_kernel, dim3(fb), dim3(1024), 0, stream, m->grad_slabs, grid + ns * L * split, m->n_params,
                       params, grad, (double)m->sc.span_eff, (double)m->sc.sp_offset, amax);
    if (hipGetLastError() != hipSuccess) return CNF_ERR_HIP;
  }
  return CNF_OK;
}

static int pass_vjp_impl(CnfModel* m, int to_base, const float* pts, const float* c, int64_t c_block,
                         const float* ybar, const float* ldbar, float* xbar, float* grad, const float* params,
                         int64_t B, void* stream_) {
  if (!m || !pts || !c || B < 0 || c_block < 1 || (!ybar && !ldbar) || (!xbar && !grad)) return CNF_ERR_INVALID;
  if (grad && !params) return CNF_ERR_INVALID;
  if (!m->params_set) return CNF_ERR_INVALID;
  if (!cnf_grad_supported(&m->cfg)) return CNF_ERR_UNSUPPORTED;
  if (B == 0) return CNF_OK;
  if (grad && !m->grad_slabs) return CNF_ERR_INVALID;      // cnf_grad_enable first
  hipStream_t stream = (hipStream_t)stream_;
  if (wait_for_params(m, stream) != CNF_OK) return CNF_ERR_HIP;
  if (grad) {
    const int r = pass_vjp_pwl(m, to_base, pts, c, c_block, ybar, ldbar, xbar, grad, params, B, stream);
    if (r != CNF_ERR_UNSUPPORTED) return r;
  }
  VjpArgs a;
  a.m = model_args(m); a.pts = pts; a.c = c; a.ybar = ybar; a.ldbar = ldbar; a.xbar = xbar;
  a.slabs = m->grad_slabs; a.n_params = m->n_params;
  a.B = B; a.c_block = c_block; a.to_base = to_base ? 1 : 0; a.div_magic = m->div_magic;
  a.fd2 = 0; a.fd_h = 0.f; a.fd_inv_dx = 0.f; a.gbar = nullptr;
  a.r1 = a.r2 = nullptr; a.rbar1 = a.rbar2 = nullptr; a.sums = nullptr; a.inv_dt = a.coef = a.drift_a = a.loss_coef = 0.f; a.drift = -1;
  const int D = m->cfg.dim, L = m->cfg.num_layers;
  const int ts = pick_tile([&](int t) { return vjp_lds_bytes(D, L, t, grad != nullptr); });
  const size_t lds = vjp_lds_bytes(D, L, ts, grad != nullptr);
  const int64_t n_tiles = (B + ts - 1) / ts;
  const int64_t max_grid = grad ? m->grad_max_blocks * 4 / (ts / 64) : (int64_t)1 << 30;
  int64_t n_slabs = 0;
  auto launch = [&](auto kernel) -> bool {
    if (!ensure_lds(kernel, lds)) return false;
    int64_t cap = (int64_t)resident_blocks_per_cu(kernel, ts, lds) * m->num_cus;
    if (cap > max_grid) cap = max_grid;
    const int64_t grid = balanced_grid(n_tiles, cap);
    n_slabs = grid * (ts / 64);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(ts), lds, stream, a);
    return true;
  };
  if (!grad) {
    if (m->fast_math ? !launch(vjp_kernel<true, false>) : !launch(vjp_kernel<false, false>)) return CNF_ERR_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
  }
  if (m->fast_math) {
    if (D == 2 ? !launch(vjp_kernel<true, true, 2>) : !launch(vjp_kernel<true, true>)) return CNF_ERR_UNSUPPORTED;
  } else {
    if (!launch(vjp_kernel<false, true>)) return CNF_ERR_UNSUPPORTED;
  }
  if (hipGetLastError() != hipSuccess) return CNF_ERR_HIP;
  const int fb = (int)((m->n_params + 31) / 32);
  hipLaunchKernelGGL(grad_finish_kernel, dim3(fb), dim3(1024), 0, stream, m->grad_slabs, n_slabs, m->n_params, params,
                     grad, (double)m->sc.span_eff, (double)m->sc.sp_offset, (uint32_t*)nullptr);
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

extern "C" int cnf_input_vjp(CnfModel* m, int to_base, const float* pts, const float* c, int64_t c_block,
                             const float* ybar, const float* ldbar, float* xbar, int64_t B, void* stream) {
  if (!xbar) return CNF_ERR_INVALID;
  return pass_vjp_impl(m, to_base, pts, c, c_block, ybar, ldbar, xbar, nullptr, nullptr, B, stream);
}

extern "C" int cnf_kinetic_potential_vjp(CnfModel* m, const float* z, int64_t count, const float* c, int32_t S, float dt,
                                         float c_kin, int32_t subtype, float pot_a, float c_pot, double* kin,
                                         double* pot, float* grad, const float* params, float* work, void* stream_) {
  if (!m || !z || !c || !kin || (grad && !params) || !work || count < 1 || S < 1 || !(dt > 0.f)) return CNF_ERR_INVALID;
  if ((subtype >= 0) != (pot != nullptr)) return CNF_ERR_INVALID;
  if (!m->params_set || (grad && !m->grad_slabs)) return CNF_ERR_INVALID;
  {      // (pass_vjp_pwl's conditions, asked before anything is launched; grad == NULL: the terms' values alone)
    const CnfConfig& g = m->cfg;
    if (!m->use_pwl || g.dim != 2 || g.hidden_size != PWL_H || g.num_bins != GK || g.mlp_num_layers != 2 || g.periodized)
      return CNF_ERR_UNSUPPORTED;
    if (grad && (!cnf_grad_supported(&g) || !m->fast_math || !m->pwl_stats || g.num_layers > 4)) return CNF_ERR_UNSUPPORTED;
  }
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t sets = pot ? 3 : 2, ns = sets * S, B = ns * count, n = (int64_t)S * count;
  if (ns > PWL_STAT_SLICES || (count & 1) || (reinterpret_cast<uintptr_t>(work) & 15)) return CNF_ERR_UNSUPPORTED;
  if (m->use_pwl == 1 && (count < 8192 || B < 262144)) return CNF_ERR_UNSUPPORTED;      // (pass_vjp_pwl's thresholds)
  if (wait_for_params(m, stream) != CNF_OK) return CNF_ERR_HIP;
  float* r = work;
  float* rbar = grad ? work + 2 * B : nullptr;
  float* tables = nullptr;
  int rc = cnf_internal_build_tables(m, stream, c, ns, &tables);
  if (rc != CNF_OK) return rc;
  rc = cnf_internal_flow_shared(m, stream, z, c, count, ns, tables, r);
  if (rc != CNF_OK) return rc;                                   // (UNSUPPORTED: nothing but the tables was touched)
  uint32_t* amax = grad ? reinterpret_cast<uint32_t*>(m->pwl_stats) : nullptr;    // zero on entry (pass_vjp_pwl)
  if (hipMemsetAsync(kin, 0, sizeof(double) * (size_t)S, stream) != hipSuccess) return CNF_ERR_HIP;
  if (pot && hipMemsetAsync(pot, 0, sizeof(double) * (size_t)S, stream) != hipSuccess) return CNF_ERR_HIP;
  TermResidArgs a;
  a.aux = nullptr; a.auxbar = nullptr; a.count = count; a.D = 2; a.amax = amax; a.n = n;
  const int64_t blocks = (n + 1024 * RESID_PER - 1) / (1024 * RESID_PER);
  a.r = r; a.rbar = rbar; a.sums = kin; a.kind = CNF_TERM_KINETIC; a.subtype = 0; a.p0 = dt; a.loss_coef = c_kin;
  hipLaunchKernelGGL(term_residual_kernel<2>, dim3((unsigned)blocks), dim3(1024), 0, stream, a);
  if (pot) {
    a.r = r + 4 * n; a.rbar = rbar ? rbar + 4 * n : nullptr; a.sums = pot; a.kind = CNF_TERM_POTENTIAL; a.subtype = subtype; a.p0 = pot_a;
    a.loss_coef = c_pot;
    hipLaunchKernelGGL(term_residual_kernel<2>, dim3((unsigned)blocks), dim3(1024), 0, stream, a);
  }
  if (hipGetLastError() != hipSuccess) return CNF_ERR_HIP;
  if (!grad) return CNF_OK;
  rc = pass_vjp_pwl(m, 0, z, c, count, rbar, nullptr, nullptr, grad, params, B, stream, 0.0f, nullptr, tables);
  if (rc != CNF_OK) {      // the adjoint maximum was left behind for a backward that did not run
    (void)hipMemsetAsync(amax, 0, 8, stream);
    return rc == CNF_ERR_UNSUPPORTED ? CNF_ERR_HIP : rc;
  }
  return CNF_OK;
}

extern "C" int cnf_neg_logprob_vjp(CnfModel* m, const float* pts, const float* c, int64_t c_block, float loss_coef,
                                   double* sums, float* grad, const float* params, int64_t B, void* stream) {
  if (!m || !pts || !c || !sums || !grad || !params || B < 0 || c_block < 1) return CNF_ERR_INVALID;
  if (!m->params_set || !m->grad_slabs) return CNF_ERR_INVALID;
  if (!cnf_grad_supported(&m->cfg)) return CNF_ERR_UNSUPPORTED;
  if (B == 0) return CNF_OK;
  if (wait_for_params(m, (hipStream_t)stream) != CNF_OK) return CNF_ERR_HIP;
  return pass_vjp_pwl(m, 1, pts, c, c_block, nullptr, nullptr, nullptr, grad, params, B, (hipStream_t)stream, loss_coef, sums);
}

extern "C" int cnf_pass_vjp(CnfModel* m, int to_base, const float* pts, const float* c, int64_t c_block,
                            const float* ybar, const float* ldbar, float* xbar, float* grad, const float* params,
                            int64_t B, void* stream) {
  return pass_vjp_impl(m, to_base, pts, c, c_block, ybar, ldbar, xbar, grad, params, B, stream);
}

// The finite-difference launches (cnf_logprob_fd_vjp, cnf_score_fd_vjp): `pts` holds n base points, the kernel
// differentiates log_prob at their 2 D evaluation points r_i -+ dx/2 e_d (generated in the kernel).
static int fd_vjp_launch(CnfModel* m, VjpArgs& a, const float* pts, const float* c, int64_t c_block, float dx,
                         float* pts_bar, float* grad, const float* params, int64_t n, hipStream_t stream) {
  if (wait_for_params(m, stream) != CNF_OK) return CNF_ERR_HIP;
  const int D = m->cfg.dim, L = m->cfg.num_layers;
  a.m = model_args(m); a.pts = pts; a.c = c; a.ybar = nullptr; a.ldbar = nullptr; a.xbar = pts_bar;
  a.slabs = m->grad_slabs; a.n_params = m->n_params;
  a.B = n * 2 * D; a.c_block = c_block; a.to_base = 1; a.div_magic = m->div_magic;
  a.fd2 = 2 * D; a.fd_h = 0.5f * dx; a.fd_inv_dx = 1.0f / dx;
  int ts = pick_tile([&](int t) { return vjp_lds_bytes(D, L, t, true); });
  while (ts < 2 * D && ts < GTS_MAX) ts <<= 1;          // a tile holds at least one group of 2 D evaluation points
  if (ts < 2 * D) return CNF_ERR_UNSUPPORTED;
  const size_t lds = vjp_lds_bytes(D, L, ts, true);
  if (lds > 160 * 1024) return CNF_ERR_UNSUPPORTED;
  const int64_t tp = (ts / (2 * D)) * (2 * D);
  const int64_t n_tiles = (a.B + tp - 1) / tp;
  const int64_t max_grid = m->grad_max_blocks * 4 / (ts / 64);
  int64_t n_slabs = 0;
  auto launch = [&](auto kernel) -> bool {
    if (!ensure_lds(kernel, lds)) return false;
    int64_t cap = (int64_t)resident_blocks_per_cu(kernel, ts, lds) * m->num_cus;
    if (cap > max_grid) cap = max_grid;
    const int64_t grid = balanced_grid(n_tiles, cap);
    n_slabs = grid * (ts / 64);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(ts), lds, stream, a);
    return true;
  };
  if (m->fast_math) {
    if (D == 2 ? !launch(vjp_kernel<true, true, 2>) : !launch(vjp_kernel<true, true>)) return CNF_ERR_UNSUPPORTED;
  } else {
    if (!launch(vjp_kernel<false, true>)) return CNF_ERR_UNSUPPORTED;
  }
  if (hipGetLastError() != hipSuccess) return CNF_ERR_HIP;
  const int fb = (int)((m->n_params + 31) / 32);
  hipLaunchKernelGGL(grad_finish_kernel, dim3(fb), dim3(1024), 0, stream, m->grad_slabs, n_slabs, m->n_params, params,
                     grad, (double)m->sc.span_eff, (double)m->sc.sp_offset, (uint32_t*)nullptr);
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

extern "C" int cnf_logprob_fd_vjp(CnfModel* m, const float* pts, const float* c, int64_t c_block, float dx,
                                  const float* gbar, float* pts_bar, float* grad, const float* params, int64_t B,
                                  void* stream_) {
  if (!m || !pts || !c || !gbar || B < 0 || c_block < 1 || !(dx > 0.f) || (!pts_bar && !grad)) return CNF_ERR_INVALID;
  if (!grad || !params) return CNF_ERR_INVALID;       // the parameter gradient is what this entry point is for
  if (!m->params_set) return CNF_ERR_INVALID;
  if (!cnf_grad_supported(&m->cfg)) return CNF_ERR_UNSUPPORTED;
  if (B == 0) return CNF_OK;
  if (!m->grad_slabs) return CNF_ERR_INVALID;          // cnf_grad_enable first
  VjpArgs a;
  a.gbar = gbar;
  a.r1 = a.r2 = nullptr; a.rbar1 = a.rbar2 = nullptr; a.sums = nullptr; a.inv_dt = a.coef = a.drift_a = a.loss_coef = 0.f; a.drift = -1;
  return fd_vjp_launch(m, a, pts, c, c_block, dx, pts_bar, grad, params, B, (hipStream_t)stream_);
}

extern "C" int cnf_score_fd_vjp(CnfModel* m, const float* r, const float* c, int64_t count, float dt, float dx, float coef,
                                int32_t drift, float a_, float loss_coef, double* sums, float* rbar, float* grad,
                                const float* params, int64_t n, void* stream_) {
  if (!m || !r || !c || !sums || !rbar || !grad || !params || n < 0 || count < 1 || !(dt > 0.f) || !(dx > 0.f))
    return CNF_ERR_INVALID;
  if (drift != -1 && drift != CNF_DRIFT_OU) return CNF_ERR_UNSUPPORTED;      // (the coupled 2-D / 3-D fields: cnf_loss_terms_grad)
  if (!m->params_set) return CNF_ERR_INVALID;
  if (!cnf_grad_supported(&m->cfg)) return CNF_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t n_slices = (n + count - 1) / count;
  if (n_slices > 0 && hipMemsetAsync(sums, 0, sizeof(double) * (size_t)n_slices, stream) != hipSuccess) return CNF_ERR_HIP;
  if (n == 0) return CNF_OK;
  if (!m->grad_slabs) return CNF_ERR_INVALID;          // cnf_grad_enable first
  const int64_t nD = n * m->cfg.dim;
  VjpArgs a;
  a.gbar = nullptr;
  a.r1 = r; a.r2 = r + nD; a.rbar1 = rbar; a.rbar2 = rbar + nD; a.sums = sums;
  a.inv_dt = 1.0f / dt; a.coef = coef; a.drift_a = a_; a.loss_coef = loss_coef; a.drift = drift;
  return fd_vjp_launch(m, a, r + 2 * nD, c, count, dx, rbar + 2 * nD, grad, params, n, stream);
}

extern "C" int cnf_score_residual(const float* r, const float* score, int64_t n, int64_t count, int32_t D, float dt,
                                  float coef, int32_t drift, float a_, float loss_coef, double* sums, float* rbar,
                                  float* sbar, void* stream_) {
  if (!r || !score || !sums || n < 0 || count < 1 || D < 1 || !(dt > 0.f) || (rbar == nullptr) != (sbar == nullptr))
    return CNF_ERR_INVALID;
  if (drift > CNF_DRIFT_LORENZ) return CNF_ERR_INVALID;
  if ((drift == CNF_DRIFT_SMILE || drift == CNF_DRIFT_NONGRADIENT) && D != 2) return CNF_ERR_INVALID;
  if (drift == CNF_DRIFT_LORENZ && D != 3) return CNF_ERR_INVALID;
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t n_slices = (n + count - 1) / count;
  if (n_slices == 0) return CNF_OK;
  if (hipMemsetAsync(sums, 0, sizeof(double) * (size_t)n_slices, stream) != hipSuccess) return CNF_ERR_HIP;
  ResidArgs a;
  a.r = r; a.score = score; a.rbar = rbar; a.sbar = sbar; a.sums = sums; a.n = n; a.count = count;
  a.D = D; a.subtype = drift; a.inv_dt = 1.0f / dt; a.coef = coef; a.a = a_; a.loss_coef = loss_coef;
  hipLaunchKernelGGL(score_residual_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

extern "C" int cnf_term_residual(int32_t kind, const float* r, const float* aux, int64_t n, int64_t count, int32_t D,
                                 int32_t subtype, float p0, float loss_coef, double* sums, float* rbar,
                                 float* auxbar, void* stream_) {
  if (!r || !sums || n < 0 || count < 1 || D < 1) return CNF_ERR_INVALID;
  if (kind != CNF_TERM_KINETIC && kind != CNF_TERM_POTENTIAL && kind != CNF_TERM_NEG_LOGPROB) return CNF_ERR_INVALID;
  if (kind == CNF_TERM_KINETIC && !(p0 > 0.f)) return CNF_ERR_INVALID;
  if (kind == CNF_TERM_NEG_LOGPROB && !aux) return CNF_ERR_INVALID;
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t n_slices = (n + count - 1) / count;
  if (n_slices > 0 && hipMemsetAsync(sums, 0, sizeof(double) * (size_t)n_slices, stream) != hipSuccess) return CNF_ERR_HIP;
  if (n == 0) return CNF_OK;
  TermResidArgs a;
  a.r = r; a.aux = aux; a.rbar = rbar; a.auxbar = auxbar; a.sums = sums; a.n = n; a.count = count;
  a.kind = kind; a.D = D; a.subtype = subtype; a.p0 = p0; a.loss_coef = loss_coef; a.amax = nullptr;
  const int64_t blocks = (n + 1024 * cnf::RESID_PER - 1) / (1024 * cnf::RESID_PER);
  const bool vec2 = D == 2 && (reinterpret_cast<uintptr_t>(r) & 7) == 0 && (reinterpret_cast<uintptr_t>(rbar) & 7) == 0;
  if (vec2) hipLaunchKernelGGL(term_residual_kernel<2>, dim3((unsigned)blocks), dim3(1024), 0, stream, a);
  else hipLaunchKernelGGL(term_residual_kernel<0>, dim3((unsigned)blocks), dim3(1024), 0, stream, a);
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

extern "C" int cnf_rkl_residual(const float* y, const float* lp, int64_t n, int32_t D, float t, float T, float beta,
                                float loss_coef, double* sum, float* ybar, float* lpbar, void* stream_) {
  if (!y || !lp || !sum || n < 0 || D < 1 || !(T > 0.f) || !(beta > 0.f) || (ybar == nullptr) != (lpbar == nullptr))
    return CNF_ERR_INVALID;
  hipStream_t stream = (hipStream_t)stream_;
  if (hipMemsetAsync(sum, 0, sizeof(double), stream) != hipSuccess) return CNF_ERR_HIP;
  if (n == 0) return CNF_OK;
  RklArgs a;
  a.y = y; a.lp = lp; a.ybar = ybar; a.lpbar = lpbar; a.sums = sum; a.n = n; a.D = D;
  a.t = t; a.T = T; a.beta = beta; a.loss_coef = loss_coef;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(rkl_residual_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}
