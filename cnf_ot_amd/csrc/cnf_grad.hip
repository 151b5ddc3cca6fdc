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

#include <math.h>

namespace cnf {

constexpr int GK = 5;              // bins
constexpr int GP = 3 * GK + 1;     // 16 spline parameters
// One sample per lane; a tile = one workgroup's samples = blockDim.x (64, 128 or 256: the host picks the size that
// puts the most waves on a CU -- a tile's LDS working set is (L + 3 .. L + 9) x D floats per sample plus 8.7 KB of
// MFMA staging per wave, and at 1 wave per SIMD a wave issues at half rate with every latency exposed).
#define GTS ((int)blockDim.x)
constexpr int GTS_MAX = TILE;
// experiment switches (scripts/exp_backward.sh; never set by the product build):
//   CNF_BWD_VALU  the conditioner of the backward kernels on the vector ALU (the round-1 form)
//   CNF_BWD_OCC1  let the backward kernels use up to 512 registers (one wave per SIMD)
#ifdef CNF_BWD_VALU
constexpr bool BWD_MFMA = false;
#else
constexpr bool BWD_MFMA = true;
#endif
#ifdef CNF_FWD_VALU
constexpr bool FWD_MFMA = false;
#else
constexpr bool FWD_MFMA = true;
#endif
#ifdef CNF_BWD_OCC1
#define CNF_BWD_MIN_BLOCKS 1
#else
#define CNF_BWD_MIN_BLOCKS 2
#endif

struct GradArgs {
  ModelArgs m;
  CnfLossSpec spec;
  const float* pts;
  const float* t;
  double* sums;          // [n_slices] loss-term sums (value of value_and_grad)
  float* slabs;          // [gridDim.x * 4][n_params]
  int64_t n_params;
  int64_t B, n_slices, pts_slice_stride;
  float scale;           // d(total loss) / d(this term's sum)
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
    U[d * GTS + s] = e < n_el ? g[base + e] : 0.0f;
  }
}

__device__ __forceinline__ int64_t cond_prefix(int d) {      // floats of conditioners 1..d-1 (H=16, M=2, P=16)
  int64_t o = 0;
  for (int dd = 1; dd < d; ++dd) o += cond_floats(dd, 16, 2, 16);
  return o;
}

// forward of one pass with every layer input kept: St[s] is the input of step s
// (St[0] filled by the caller), St[L] the result.  Returns the log-det sum.
// The direction is a RUNTIME (wave-uniform) flag and every pass of the kernel
// goes through the single call site of this function and of pass_bwd: the
// conditioner code (the bulk) exists once, so the kernel fits the instruction
// cache (the first version inlined 10 forward + 6 backward copies: 34 k
// instructions, 4x the cache).
template <bool FAST>
__device__ __forceinline__ float pass_fwd_stash(const ModelArgs& a, const float* tab, float* St, float c,
                                                bool to_base) {
  const int D = a.D;
  uniform_ptr weights = as_uniform(a.prep + hdr_floats(GK));
  float acc = 0.0f;
  for (int s = 0; s < a.L; ++s) {
    const int l = to_base ? a.L - 1 - s : s;
    const bool odd = l & 1;
    const int first_idx = odd ? D - 1 : 0, idx_step = odd ? -1 : 1;
    const float* cu = St + s * D * GTS + threadIdx.x;
    float* co = St + (s + 1) * D * GTS + threadIdx.x;
    float o, ld;
    if (to_base) table_spline<GK, false, FAST, float>(tab, cu[first_idx * GTS], a.sc, o, ld);
    else table_spline<GK, true, FAST, float>(tab, cu[first_idx * GTS], a.sc, o, ld);
    co[first_idx * GTS] = o;
    acc += ld;
    uniform_ptr w = weights + l * a.per_layer;
    [[maybe_unused]] const float* wq = a.wq + l * a.per_layer_q;
    for (int d = 1; d < D; ++d) {
      const int i = first_idx + d * idx_step;
      float th[GP];
      if constexpr (FAST && BWD_MFMA && FWD_MFMA) {        // matrix cores (see cnf_backward.h); `wq` walks the MFMA-layout weights
        conditioner_mfma_lane1(reinterpret_cast<const f4*>(wq), w, d, c, to_base ? co : cu, first_idx, idx_step, GTS, th);
        wq += cond_floats_mfma(d, 2);
      } else {
        conditioner<16, GP, float>(w, d, 2, c, to_base ? co : cu, first_idx, idx_step, GTS, th);
      }
      if (to_base) cond_spline<GK, false, FAST, float>(th, cu[i * GTS], a.sc, o, ld);
      else cond_spline<GK, true, FAST, float>(th, cu[i * GTS], a.sc, o, ld);
      co[i * GTS] = o;
      acc += ld;
      w += cond_floats(d, 16, 2, GP);
    }
  }
  return acc;
}

// backward of the pass whose stash is in St.  Aa holds the adjoint of the final
// output on entry; the function ping-pongs between Aa and Ab and returns the
// buffer that holds the adjoint of the pass input.
template <bool FAST, bool WGRAD = true>
__device__ __forceinline__ float* pass_bwd(const ModelArgs& a, const float* tab, const float* St, float* Aa,
                                           float* Ab, float ld_bar, float c, bool to_base, float* gslab,
                                           float* stage, FirstAcc& fa) {
  const int D = a.D;
  uniform_ptr weights = as_uniform(a.prep + hdr_floats(GK));
  float* Ao = Aa;
  float* Au = Ab;
  for (int s = a.L - 1; s >= 0; --s) {
    const int l = to_base ? a.L - 1 - s : s;
    const bool odd = l & 1;
    const int first_idx = odd ? D - 1 : 0, idx_step = odd ? -1 : 1;
    const float* cu = St + s * D * GTS + threadIdx.x;
    const float* co = St + (s + 1) * D * GTS + threadIdx.x;
    float* ao = Ao + threadIdx.x;
    float* au = Au + threadIdx.x;
    for (int d = 0; d < D; ++d) au[d * GTS] = 0.0f;
    int64_t off = cond_prefix(D);                     // end of this layer's conditioners
    [[maybe_unused]] int64_t offq = 0;                 // the same in the MFMA-layout weights
    if constexpr (FAST && BWD_MFMA) { for (int dd = 1; dd < D; ++dd) offq += cond_floats_mfma(dd, 2); }
    for (int d = D - 1; d >= 1; --d) {
      off -= cond_floats(d, 16, 2, GP);
      const int i = first_idx + d * idx_step;
      uniform_ptr w = weights + l * a.per_layer + off;
      float* gw = WGRAD ? gslab + GP + l * a.per_layer + off : nullptr;
      WgradPre pre;
      float th[GP], tb[GP];
      if constexpr (FAST && BWD_MFMA) {      // recompute and data backprop on the matrix cores (cnf_backward.h)
        offq -= cond_floats_mfma(d, 2);
        uint32_t mask1;
        float h2m[4][4];
        conditioner_mfma_keep(reinterpret_cast<const f4*>(a.wq + l * a.per_layer_q + offq), w, d, c, to_base ? co : cu,
                              first_idx, idx_step, GTS, stage, mask1, h2m, th);
        // the accumulator tiles: fetched here, ~300 instructions (the spline backward) ahead of their first use;
        // any earlier and the 15 registers they occupy push the kernel past 256 (2 waves per SIMD)
        if constexpr (WGRAD) pre = wgrad_prefetch(gw, d);
        float vb;
        if (to_base) vb = cond_spline_bwd<GK, false, FAST>(th, cu[i * GTS], co[i * GTS], ao[i * GTS], ld_bar, a.sc, tb);
        else vb = cond_spline_bwd<GK, true, FAST>(th, cu[i * GTS], co[i * GTS], ao[i * GTS], ld_bar, a.sc, tb);
        au[i * GTS] += vb;
        conditioner_bwd_mfma<WGRAD>(a.prep + hdr_floats(GK) + l * a.per_layer + off, w, d, c, to_base ? co : cu,
                                    first_idx, idx_step, GTS, mask1, h2m, tb, to_base ? ao : au, gw, stage, pre);
      } else {
        if constexpr (WGRAD) pre = wgrad_prefetch(gw, d);
        float h1[16], h2[16];
        conditioner_keep(w, d, c, to_base ? co : cu, first_idx, idx_step, GTS, h1, h2, th);
        float vb;
        if (to_base) vb = cond_spline_bwd<GK, false, FAST>(th, cu[i * GTS], co[i * GTS], ao[i * GTS], ld_bar, a.sc, tb);
        else vb = cond_spline_bwd<GK, true, FAST>(th, cu[i * GTS], co[i * GTS], ao[i * GTS], ld_bar, a.sc, tb);
        au[i * GTS] += vb;
        conditioner_bwd<WGRAD>(w, d, c, to_base ? co : cu, first_idx, idx_step, GTS, h1, h2, tb, to_base ? ao : au,
                               gw, stage ? stage + 16 * STG : nullptr, pre);
      }
    }
    float vb0;
    if (to_base) vb0 = table_spline_bwd<GK, false>(tab, cu[first_idx * GTS], co[first_idx * GTS], ao[first_idx * GTS],
                                                   ld_bar, a.sc, fa.Wb, fa.Hb, fa.Db);
    else vb0 = table_spline_bwd<GK, true>(tab, cu[first_idx * GTS], co[first_idx * GTS], ao[first_idx * GTS],
                                          ld_bar, a.sc, fa.Wb, fa.Hb, fa.Db);
    au[first_idx * GTS] += vb0;
    float* t = Ao; Ao = Au; Au = t;
  }
  return Ao;
}

__device__ __forceinline__ float base_lp(const float* col, int D) {
  float b = 0.0f;
  for (int d = 0; d < D; ++d) { const float x = col[d * GTS]; b = fmaf(-0.5f * x, x, b); }
  return b - (float)(D * HALF_LOG_2PI);
}

// R3b[e] -= sum_d ubar_d * d drift_d / d r_e   (flow_matching_loss_fn's target field)
__device__ __forceinline__ void drift_vjp(const float* r3, const float* ub, float* r3b, int D, int subtype, float a) {
  switch (subtype) {
    case CNF_DRIFT_SMILE: {
      const float x = r3[0], y = r3[GTS], q = x * x + y * y - 4.0f, u0 = ub[0], u1 = ub[GTS];
      r3b[0] -= u0 * (-a * (q + 2.0f * x * x)) + u1 * (-a * 2.0f * x * y);
      r3b[GTS] -= u0 * (-a * 2.0f * x * y) + u1 * (-a * (q + 2.0f * y * y + 2.0f));
      break;
    }
    case CNF_DRIFT_NONGRADIENT: {
      const float u0 = ub[0], u1 = ub[GTS];
      r3b[0] -= u0 * (-a) + u1 * 0.5f;
      r3b[GTS] -= u0 * (-0.5f) + u1 * (-a);
      break;
    }
    case CNF_DRIFT_LORENZ: {
      const float x = r3[0], y = r3[GTS], z = r3[2 * GTS], u0 = ub[0], u1 = ub[GTS], u2 = ub[2 * GTS];
      r3b[0] -= u0 * -10.0f + u1 * (28.0f - 9.0f * z) + u2 * 9.0f * y;
      r3b[GTS] -= u0 * 10.0f - u1 + u2 * 9.0f * x;
      r3b[2 * GTS] -= u1 * (-9.0f * x) + u2 * (-8.0f / 3.0f);
      break;
    }
    default:
      for (int d = 0; d < D; ++d) r3b[d * GTS] += a * ub[d * GTS];
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

template <bool FAST>
__global__ __launch_bounds__(GTS_MAX, CNF_BWD_MIN_BLOCKS) void grad_kernel(const GradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int HDR = hdr_floats(GK);
  const int D = a.m.D, L = a.m.L;
  const int DT = D * GTS;
  float* tab = lds;
  float* Nn = lds + HDR;
  float* St = Nn + DT;                  // (L+1) stashes
  float* Aa = St + (L + 1) * DT;
  float* Ab = Aa + DT;
  float* V = Ab + DT;
  float* R3 = V + DT;
  float* R3b = R3 + DT;
  float* Ub = R3b + DT;
  float* stage = Ub + DT + (threadIdx.x >> 6) * STAGE_FLOATS;
  for (int i = threadIdx.x; i < HDR; i += GTS) tab[i] = a.m.prep[i];
  const int tid = threadIdx.x;
  const int kind = a.spec.kind;
  float* gslab = a.slabs + ((int64_t)blockIdx.x * (GTS >> 6) + (tid >> 6)) * a.n_params;
  slab_clear(gslab, a.n_params, tid & 63);
  FirstAcc fa;
#pragma unroll
  for (int j = 0; j < GK; ++j) { fa.Wb[j] = 0.0f; fa.Hb[j] = 0.0f; }
#pragma unroll
  for (int j = 0; j <= GK; ++j) fa.Db[j] = 0.0f;

  const float dt = a.spec.dt, dx = a.spec.dx, coef = a.spec.coef;
  const bool score = kind == CNF_TERM_KINETIC_SCORE || kind == CNF_TERM_FLOW_MATCHING;
  // number of steps of the pass program
  const int n_steps = kind == CNF_TERM_KINETIC ? 3 : (score ? 3 + 3 * D + 3 : 1);

  const int64_t tiles_per_slice = (a.B + GTS - 1) / GTS;
  const int64_t n_tiles = tiles_per_slice * a.n_slices;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t slice = tile / tiles_per_slice;
    const int64_t tile_start = (tile - slice * tiles_per_slice) * GTS;
    const bool valid = tile_start + tid < a.B;
    const float sc = valid ? a.scale : 0.0f;
    __syncthreads();
    tile_load1(a.pts + slice * a.pts_slice_stride * D, Nn, D, a.div_magic, tile_start, a.B);
    const float t = a.t[slice];
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
        for (int e = 0; e < D; ++e) s0[e * GTS] = r3[e * GTS];
        s0[dd * GTS] += role == R_LPM ? -0.5f * dx : 0.5f * dx;
      } else {
        for (int e = 0; e < D; ++e) s0[e * GTS] = n_[e * GTS];
      }
      const float ldsum = pass_fwd_stash<FAST>(a.m, tab, St, c, to_base);
      // ---- act on the result, prepare the seeds
      bool do_bwd = false;
      float ld_bar = 0.0f;
      switch (role) {
        case R_NEG: {
          lossv = -(base_lp(sL, D) + ldsum);
          for (int e = 0; e < D; ++e) aa[e * GTS] = sc * sL[e * GTS];      // d(-lp)/dx_e = x_e
          ld_bar = -sc; do_bwd = true;
          break;
        }
        case R_POT: {
          const float pa = a.spec.a;
          if (a.spec.subtype == CNF_POT_DOUBLE_WELL) {
            float sm = 0.0f, sp = 0.0f;
            for (int e = 0; e < D; ++e) { const float r = sL[e * GTS]; sm = fmaf(r - pa, r - pa, sm); sp = fmaf(r + pa, r + pa, sp); }
            lossv = sm * sp * 0.25f;
            for (int e = 0; e < D; ++e) { const float r = sL[e * GTS]; aa[e * GTS] = sc * 0.5f * ((r - pa) * sp + (r + pa) * sm); }
          } else {
            float s2 = 0.0f;
            for (int e = 0; e < D; ++e) { const float r = sL[e * GTS]; s2 = fmaf(r, r, s2); }
            if (a.spec.subtype == CNF_POT_OBSTACLE) {
              lossv = 50.0f * expf(-0.5f * s2);
              for (int e = 0; e < D; ++e) aa[e * GTS] = -sc * lossv * sL[e * GTS];
            } else {
              lossv = 0.5f * s2;
              for (int e = 0; e < D; ++e) aa[e * GTS] = sc * sL[e * GTS];
            }
          }
          do_bwd = true;
          break;
        }
        case R_RKL: {
          const float lp = base_lp(n_, D) - ldsum;
          float s2 = 0.0f;
          for (int e = 0; e < D; ++e) { const float r = sL[e * GTS]; s2 = fmaf(r, r, s2); }
          const float Tt = a.spec.T, vs = 2.0f / a.spec.beta * (Tt + 1.0f), vt = 2.0f / a.spec.beta;
          const float ws = (Tt - t) / Tt, wt = t / Tt;
          const float ls = -0.5f * D * logf(6.283185307179586f * vs), lt = -0.5f * D * logf(6.283185307179586f * vt);
          const float as = -0.5f * s2 / vs + ls, at = -0.5f * s2 / vt + lt;
          const float mx = fmaxf(as, at);
          const float es = expf(as - mx) * ws, et = expf(at - mx) * wt;
          lossv = lp - (mx + logf(es + et));
          const float g = (es / vs + et / vt) / (es + et);      // -d logmix / d y_e = g * y_e
          for (int e = 0; e < D; ++e) aa[e * GTS] = sc * g * sL[e * GTS];
          ld_bar = -sc; do_bwd = true;
          break;
        }
        case R_R1:
          for (int e = 0; e < D; ++e) v_[e * GTS] = sL[e * GTS];
          break;
        case R_R2: {
          const float inv_dt = 1.0f / dt;
          for (int e = 0; e < D; ++e) v_[e * GTS] = (sL[e * GTS] - v_[e * GTS]) * inv_dt;      // velocity
          if (kind == CNF_TERM_KINETIC) {
            for (int e = 0; e < D; ++e) {
              const float v = v_[e * GTS];
              lossv = fmaf(v, v, lossv);
              ub[e * GTS] = 2.0f * sc * v;
              aa[e * GTS] = ub[e * GTS] * inv_dt;
            }
            do_bwd = true;                          // the r2 stash is live
          }
          break;
        }
        case R_R3:
          for (int e = 0; e < D; ++e) { r3[e * GTS] = sL[e * GTS]; r3b[e * GTS] = 0.0f; }
          break;
        case R_LPP:
          lp_plus = base_lp(sL, D) + ldsum;
          break;
        case R_LPM: {
          const float lp_minus = base_lp(sL, D) + ldsum;
          float u = fmaf((lp_plus - lp_minus) / dx, coef, v_[dd * GTS]);
          if (kind == CNF_TERM_FLOW_MATCHING) u -= drift_of<float>(r3, dd, D, GTS, a.spec.subtype, a.spec.a);
          lossv = fmaf(u, u, lossv);
          ubar = 2.0f * sc * u;
          ub[dd * GTS] = ubar;
          ld_bar = -ubar * coef / dx;                // d u / d lp_minus
          for (int e = 0; e < D; ++e) aa[e * GTS] = -ld_bar * sL[e * GTS];     // d base / d x_e = -x_e
          do_bwd = true;
          break;
        }
        case R_LPPB:
          ld_bar = ubar * coef / dx;
          for (int e = 0; e < D; ++e) aa[e * GTS] = -ld_bar * sL[e * GTS];
          do_bwd = true;
          break;
        case R_R3B:
          if (kind == CNF_TERM_FLOW_MATCHING) drift_vjp(r3, ub, r3b, D, a.spec.subtype, a.spec.a);
          for (int e = 0; e < D; ++e) aa[e * GTS] = r3b[e * GTS];
          do_bwd = true;
          break;
        case R_R2B:
          for (int e = 0; e < D; ++e) aa[e * GTS] = ub[e * GTS] / dt;
          do_bwd = true;
          break;
        default:   // R_R1B
          for (int e = 0; e < D; ++e) aa[e * GTS] = -ub[e * GTS] / dt;
          do_bwd = true;
          break;
      }
      if (do_bwd) {
        const float* ain = pass_bwd<FAST>(a.m, tab, St, Aa, Ab, ld_bar, c, to_base, gslab, stage, fa) + tid;
        if (role == R_LPM || role == R_LPPB)
          for (int e = 0; e < D; ++e) r3b[e * GTS] += ain[e * GTS];
      }
    }
    float part = valid ? lossv : 0.0f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if ((tid & 63) == 0) atomicAdd(a.sums + slice, (double)part);
  }
  // per-bin adjoint sums of the shared `first` spline: wave reduce, one owner write
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
};

// WGRAD=true additionally accumulates the parameter gradient of the pass (the
// backward of a differentiable flow op: cnf_pass_vjp).
template <bool FAST, bool WGRAD>
__global__ __launch_bounds__(GTS_MAX, CNF_BWD_MIN_BLOCKS) void vjp_kernel(const VjpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int HDR = hdr_floats(GK);
  const int D = a.m.D, L = a.m.L, DT = D * GTS;
  float* tab = lds;
  float* St = lds + HDR;
  float* Aa = St + (L + 1) * DT;
  float* Ab = Aa + DT;
  float* stage = Ab + DT + (threadIdx.x >> 6) * STAGE_FLOATS;       // (WGRAD = false: only the h1 region is touched)
  float* gslab = WGRAD ? a.slabs + ((int64_t)blockIdx.x * (GTS >> 6) + (threadIdx.x >> 6)) * a.n_params : nullptr;
  if (WGRAD) slab_clear(gslab, a.n_params, threadIdx.x & 63);
  for (int i = threadIdx.x; i < HDR; i += GTS) tab[i] = a.m.prep[i];
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
        St[e * GTS + tid] = v;
        Aa[e * GTS + tid] = 0.0f;
      }
      const int64_t n_base = a.B / a.fd2;
      c = valid ? a.c[a.c_block >= n_base ? 0 : ib / a.c_block] : 0.0f;
      ld_bar = valid ? ((k & 1) ? -a.gbar[ib * D + dd] : a.gbar[ib * D + dd]) * a.fd_inv_dx : 0.0f;
    } else {
      tile_load1(a.pts, St, D, a.div_magic, tile_start, a.B);
      if (a.ybar) tile_load1(a.ybar, Aa, D, a.div_magic, tile_start, a.B);
      else for (int e = tid; e < DT; e += GTS) Aa[e] = 0.0f;
      c = i < a.B ? a.c[a.c_block >= a.B ? 0 : i / a.c_block] : 0.0f;
      ld_bar = (a.ldbar && i < a.B) ? a.ldbar[i] : 0.0f;
    }
    __syncthreads();
    pass_fwd_stash<FAST>(a.m, tab, St, c, a.to_base != 0);
    if (a.fd2) {       // log_prob = sum -x^2/2 + ildj: the adjoint of the recovered base point is -ld_bar x
      const float* sL = St + L * DT + tid;
      for (int e = 0; e < D; ++e) Aa[e * GTS + tid] = -ld_bar * sL[e * GTS];
    }
    float* ain = pass_bwd<FAST, WGRAD>(a.m, tab, St, Aa, Ab, ld_bar, c, a.to_base != 0, gslab, stage, fa);
    __syncthreads();
    if (a.fd2) {       // xbar[i, e] = sum over the fd2 evaluation points of base point i (fixed order: deterministic)
      if (a.xbar) {
        const int groups = tp / a.fd2;
        const int64_t first_base = tile_start / a.fd2, n_base = a.B / a.fd2;
        for (int idx = tid; idx < groups * D; idx += GTS) {
          const int il = idx / D, e = idx - il * D;
          if (first_base + il < n_base) {
            float sum = 0.0f;
            for (int k = 0; k < a.fd2; ++k) sum += ain[e * GTS + il * a.fd2 + k];
            a.xbar[(first_base + il) * D + e] = sum;
          }
        }
      }
    } else if (a.xbar) {      // coalesced store of the input adjoints
      const int64_t base = tile_start * D;
      const int n_el = (int)(a.B - tile_start < GTS ? a.B - tile_start : GTS) * D;
      for (int e = tid; e < GTS * D; e += GTS) {
        const int s = a.div_magic ? (int)__umulhi((uint32_t)e, a.div_magic) : e, d = e - s * D;
        if (e < n_el) a.xbar[base + e] = ain[d * GTS + s];
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

// grad[p] += sum over slabs; the first 16 entries are per-bin adjoint sums of
// the `first` spline and go through the softmax / softplus Jacobians (float64).
// Block = 32 parameters x 32 slab stripes (a thread that walks all slabs alone
// pays one dependent L2 round trip per slab: 1-2 ms for 2 048 slabs).
__global__ __launch_bounds__(1024) void grad_finish_kernel(const float* __restrict__ slabs, int64_t n_slabs,
                                                           int64_t n_params, const float* __restrict__ params,
                                                           float* __restrict__ grad, double span_eff,
                                                           double sp_offset) {
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
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  float acc = 0.0f;
  if (i < a.n) {
    const int D = a.D;
    float s2 = 0.0f;
    for (int d = 0; d < D; ++d) { const float r = a.y[i * D + d]; s2 = fmaf(r, r, s2); }
    const float vs = 2.0f / a.beta * (a.T + 1.0f), vt = 2.0f / a.beta;
    const float ws = (a.T - a.t) / a.T, wt = a.t / a.T;
    const float ls = -0.5f * D * logf(6.283185307179586f * vs), lt = -0.5f * D * logf(6.283185307179586f * vt);
    const float as = -0.5f * s2 / vs + ls, at = -0.5f * s2 / vt + lt;
    const float mx = fmaxf(as, at);
    const float es = expf(as - mx) * ws, et = expf(at - mx) * wt;
    acc = a.lp[i] - (mx + logf(es + et));
    if (a.ybar) {
      const float g = (es / vs + et / vt) / (es + et);      // -d logmix / d y_e = g * y_e
      for (int d = 0; d < D; ++d) a.ybar[i * D + d] = a.loss_coef * g * a.y[i * D + d];
      a.lpbar[i] = a.loss_coef;
    }
  }
  float part = acc;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(a.sums, (double)part);
}

}  // namespace cnf

using namespace cnf;

#undef GTS
static size_t grad_lds_bytes(int D, int L, int ts = GTS_MAX) {
  // tab + noise + (L+1) stashes + 2 adjoint buffers + velocity, r3, r3_bar, u_bar + one MFMA staging area per wave
  return (size_t)(hdr_floats(GK) + D * ts * (1 + (L + 1) + 6) + (ts / 64) * STAGE_FLOATS) * sizeof(float);
}
static size_t vjp_lds_bytes(int D, int L, int ts, bool wgrad) {
  (void)wgrad;       // the MFMA recompute stages h1 even without weight gradients
  return (size_t)(hdr_floats(GK) + D * ts * ((L + 1) + 2) + (ts / 64) * STAGE_FLOATS) * sizeof(float);
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

extern "C" int cnf_loss_terms_grad(CnfModel* m, const CnfLossSpec* spec, const float* pts, int pts_shared,
                                   const float* t, int64_t n_slices, int64_t B, float scale, double* sums,
                                   float* grad, const float* params, void* stream_) {
  if (!m || !spec || !pts || !t || !sums || !grad || !params || n_slices < 0 || B < 0) return CNF_ERR_INVALID;
  if (!m->params_set) return CNF_ERR_INVALID;
  if (!cnf_grad_supported(&m->cfg)) return CNF_ERR_UNSUPPORTED;
  if (spec->kind < CNF_TERM_KINETIC || spec->kind > CNF_TERM_NEG_LOGPROB) return CNF_ERR_INVALID;
  const int D = m->cfg.dim, L = m->cfg.num_layers;
  if (spec->kind <= CNF_TERM_FLOW_MATCHING && !(spec->dt > 0.f)) return CNF_ERR_INVALID;
  if ((spec->kind == CNF_TERM_KINETIC_SCORE || spec->kind == CNF_TERM_FLOW_MATCHING) && !(spec->dx > 0.f)) return CNF_ERR_INVALID;
  if (spec->kind == CNF_TERM_FLOW_MATCHING) {
    if ((spec->subtype == CNF_DRIFT_SMILE || spec->subtype == CNF_DRIFT_NONGRADIENT) && D != 2) return CNF_ERR_INVALID;
    if (spec->subtype == CNF_DRIFT_LORENZ && D != 3) return CNF_ERR_INVALID;
  }
  hipStream_t stream = (hipStream_t)stream_;
  if (n_slices == 0) return CNF_OK;
  if (hipMemsetAsync(sums, 0, sizeof(double) * (size_t)n_slices, stream) != hipSuccess) return CNF_ERR_HIP;
  if (B == 0) return CNF_OK;
  if (!m->grad_slabs) return CNF_ERR_INVALID;      // cnf_grad_enable first
  if (wait_for_params(m, stream) != CNF_OK) return CNF_ERR_HIP;

  GradArgs a;
  a.m = model_args(m); a.spec = *spec; a.pts = pts; a.t = t; a.sums = sums;
  a.slabs = m->grad_slabs; a.n_params = m->n_params;
  a.B = B; a.n_slices = n_slices; a.pts_slice_stride = pts_shared ? 0 : B;
  a.scale = scale; a.div_magic = m->div_magic;
  const int ts = pick_tile([&](int t) { return grad_lds_bytes(D, L, t); });
  int64_t grid = ((B + ts - 1) / ts) * n_slices;
  if (grid > m->grad_max_blocks * 4 / (ts / 64)) grid = m->grad_max_blocks * 4 / (ts / 64);
  const size_t lds = grad_lds_bytes(D, L, ts);
  if (lds > 160 * 1024) return CNF_ERR_UNSUPPORTED;
  const int64_t n_slabs = grid * (ts / 64);
  if (m->fast_math) {
    if (!ensure_lds(grad_kernel<true>, lds)) return CNF_ERR_HIP;
    hipLaunchKernelGGL(grad_kernel<true>, dim3((unsigned)grid), dim3(ts), lds, stream, a);
  } else {
    if (!ensure_lds(grad_kernel<false>, lds)) return CNF_ERR_HIP;
    hipLaunchKernelGGL(grad_kernel<false>, dim3((unsigned)grid), dim3(ts), lds, stream, a);
  }
  if (hipGetLastError() != hipSuccess) return CNF_ERR_HIP;
  const int fb = (int)((m->n_params + 31) / 32);
  hipLaunchKernelGGL(grad_finish_kernel, dim3(fb), dim3(1024), 0, stream, m->grad_slabs, n_slabs, m->n_params, params,
                     grad, (double)m->sc.span_eff, (double)m->sc.sp_offset);
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

extern "C" int cnf_grad_enable(CnfModel* m, int64_t max_blocks) {
  if (!m) return CNF_ERR_INVALID;
  if (!cnf_grad_supported(&m->cfg)) return CNF_ERR_UNSUPPORTED;
  if (max_blocks <= 0) max_blocks = (int64_t)m->num_cus * 2;
  if (m->grad_slabs && m->grad_max_blocks >= max_blocks) return CNF_OK;
  if (m->grad_slabs) { (void)hipFree(m->grad_slabs); m->grad_slabs = nullptr; }
  if (hipMalloc((void**)&m->grad_slabs, sizeof(float) * (size_t)(max_blocks * 4 * m->n_params)) != hipSuccess) return CNF_ERR_NOMEM;
  m->grad_max_blocks = max_blocks;
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
  VjpArgs a;
  a.m = model_args(m); a.pts = pts; a.c = c; a.ybar = ybar; a.ldbar = ldbar; a.xbar = xbar;
  a.slabs = m->grad_slabs; a.n_params = m->n_params;
  a.B = B; a.c_block = c_block; a.to_base = to_base ? 1 : 0; a.div_magic = m->div_magic;
  a.fd2 = 0; a.fd_h = 0.f; a.fd_inv_dx = 0.f; a.gbar = nullptr;
  const int D = m->cfg.dim, L = m->cfg.num_layers;
  const int ts = pick_tile([&](int t) { return vjp_lds_bytes(D, L, t, grad != nullptr); });
  size_t lds = vjp_lds_bytes(D, L, ts, grad != nullptr);
  int64_t grid = (B + ts - 1) / ts;
  if (!grad) {
    if (grid > (int64_t)m->num_cus * 4 * (GTS_MAX / ts)) grid = (int64_t)m->num_cus * 4 * (GTS_MAX / ts);
    if (m->fast_math) {
      if (!ensure_lds(vjp_kernel<true, false>, lds)) return CNF_ERR_UNSUPPORTED;
      hipLaunchKernelGGL((vjp_kernel<true, false>), dim3((unsigned)grid), dim3(ts), lds, stream, a);
    } else {
      if (!ensure_lds(vjp_kernel<false, false>, lds)) return CNF_ERR_UNSUPPORTED;
      hipLaunchKernelGGL((vjp_kernel<false, false>), dim3((unsigned)grid), dim3(ts), lds, stream, a);
    }
    return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
  }
  if (grid > m->grad_max_blocks * 4 / (ts / 64)) grid = m->grad_max_blocks * 4 / (ts / 64);
  const int64_t n_slabs = grid * (ts / 64);
  if (m->fast_math) {
    if (!ensure_lds(vjp_kernel<true, true>, lds)) return CNF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((vjp_kernel<true, true>), dim3((unsigned)grid), dim3(ts), lds, stream, a);
  } else {
    if (!ensure_lds(vjp_kernel<false, true>, lds)) return CNF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((vjp_kernel<false, true>), dim3((unsigned)grid), dim3(ts), lds, stream, a);
  }
  if (hipGetLastError() != hipSuccess) return CNF_ERR_HIP;
  const int fb = (int)((m->n_params + 31) / 32);
  hipLaunchKernelGGL(grad_finish_kernel, dim3(fb), dim3(1024), 0, stream, m->grad_slabs, n_slabs, m->n_params, params,
                     grad, (double)m->sc.span_eff, (double)m->sc.sp_offset);
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

extern "C" int cnf_input_vjp(CnfModel* m, int to_base, const float* pts, const float* c, int64_t c_block,
                             const float* ybar, const float* ldbar, float* xbar, int64_t B, void* stream) {
  if (!xbar) return CNF_ERR_INVALID;
  return pass_vjp_impl(m, to_base, pts, c, c_block, ybar, ldbar, xbar, nullptr, nullptr, B, stream);
}

extern "C" int cnf_pass_vjp(CnfModel* m, int to_base, const float* pts, const float* c, int64_t c_block,
                            const float* ybar, const float* ldbar, float* xbar, float* grad, const float* params,
                            int64_t B, void* stream) {
  return pass_vjp_impl(m, to_base, pts, c, c_block, ybar, ldbar, xbar, grad, params, B, stream);
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
  hipStream_t stream = (hipStream_t)stream_;
  if (wait_for_params(m, stream) != CNF_OK) return CNF_ERR_HIP;
  const int D = m->cfg.dim, L = m->cfg.num_layers;
  VjpArgs a;
  a.m = model_args(m); a.pts = pts; a.c = c; a.ybar = nullptr; a.ldbar = nullptr; a.xbar = pts_bar;
  a.slabs = m->grad_slabs; a.n_params = m->n_params;
  a.B = B * 2 * D; a.c_block = c_block; a.to_base = 1; a.div_magic = m->div_magic;
  a.fd2 = 2 * D; a.fd_h = 0.5f * dx; a.fd_inv_dx = 1.0f / dx; a.gbar = gbar;
  int ts = pick_tile([&](int t) { return vjp_lds_bytes(D, L, t, true); });
  while (ts < 2 * D && ts < GTS_MAX) ts <<= 1;          // a tile holds at least one group of 2 D evaluation points
  if (ts < 2 * D) return CNF_ERR_UNSUPPORTED;
  const size_t lds = vjp_lds_bytes(D, L, ts, true);
  if (lds > 160 * 1024) return CNF_ERR_UNSUPPORTED;
  const int64_t tp = (ts / (2 * D)) * (2 * D);
  int64_t grid = (a.B + tp - 1) / tp;
  if (grid > m->grad_max_blocks * 4 / (ts / 64)) grid = m->grad_max_blocks * 4 / (ts / 64);
  const int64_t n_slabs = grid * (ts / 64);
  if (m->fast_math) {
    if (!ensure_lds(vjp_kernel<true, true>, lds)) return CNF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((vjp_kernel<true, true>), dim3((unsigned)grid), dim3(ts), lds, stream, a);
  } else {
    if (!ensure_lds(vjp_kernel<false, true>, lds)) return CNF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((vjp_kernel<false, true>), dim3((unsigned)grid), dim3(ts), lds, stream, a);
  }
  if (hipGetLastError() != hipSuccess) return CNF_ERR_HIP;
  const int fb = (int)((m->n_params + 31) / 32);
  hipLaunchKernelGGL(grad_finish_kernel, dim3(fb), dim3(1024), 0, stream, m->grad_slabs, n_slabs, m->n_params, params,
                     grad, (double)m->sc.span_eff, (double)m->sc.sp_offset);
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
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
  hipLaunchKernelGGL(rkl_residual_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}
