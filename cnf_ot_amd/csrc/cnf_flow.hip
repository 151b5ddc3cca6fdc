// cnf_flow.hip -- gfx950 kernels and C ABI of the conditional RQS flow engine.
// Declarations and the reference interfaces they replace: include/cnf_ot_amd.h.
//
// Kernel design (DESIGN.md has the numbers):
//  * one sample per lane, a 256-sample tile per workgroup, grid-stride over
//    tiles; the tile's [256, D] rows are one contiguous HBM range, so loads and
//    stores are fully coalesced and transposed through LDS into per-thread
//    columns u[d][tid] (conflict-free per-lane access);
//  * all L layers, all D dimensions, conditioner MLP + spline + log|det J|
//    accumulate are fused: HBM sees only the input row, the output row and one
//    float of log-det / log-prob per sample;
//  * conditioner weights are wave-uniform: they reach the per-lane FMAs as
//    scalar (SGPR) operands through the scalar cache, not through LDS/VGPRs;
//  * the shared `first` spline is pre-normalised (float64, once per parameter
//    set) into a 12-float-per-bin table that is staged in LDS and gathered by
//    per-lane bin index.
#include "cnf_common.h"
#include <cstdlib>
#include "cnf_pwl_build.h"

#include <math.h>
#include <new>
#include <stdlib.h>
#include <string.h>

namespace cnf {


enum CMode { C_SINGLE = 0, C_PER_SAMPLE = 1, C_TILE_UNIFORM = 2, C_GENERIC = 3 };
enum AuxMode { AUX_LOGDET = 0, AUX_LOGPROB = 1 };


template <class R> struct FlowArgsT {
  ModelArgs m;
  const R* in;           // [B, D]
  const R* c;            // conditions
  R* out;                // [B, D] or null
  R* aux;                // [B] logdet / logprob, or null
  int64_t B;
  int64_t c_block;
  int32_t c_mode, aux_mode;
  int32_t div_magic;     // ceil(2^32 / D): e / D == umulhi(e, magic) for e < 2^16
  // device-side choice between two kernels enqueued for the same call (uniform-condition detection,
  // cond_uniform_kernel): this kernel runs only if (*gate == gate_epoch) == gate_want; null: always
  const uint32_t* gate;
  uint32_t gate_epoch;
  int32_t gate_want;
  // finite-difference mode (cnf_logprob_fd; data -> base only): `in` holds B / fd2 points r_i and evaluation
  // point j = i * fd2 + 2 d + s is r_i + (s ? -fd_h : +fd_h) e_d (fd2 = 2 D); aux[i * D + d] receives
  // (log_prob(j) - log_prob(j + 1)) * fd_inv_dx.  fd2 = 0: off.
  int32_t fd2;
  R fd_h, fd_inv_dx;
  // in == null (float32, base -> data only; cnf_sample_logprob_seeded): the points are base noise drawn in the kernel,
  // sample i = stream sample first_sample + (i / c_block) * slice_stride + i % c_block of the cnf_fill_normal stream
  uint64_t seed;
  int64_t first_sample, slice_stride;
};
typedef FlowArgsT<float> FlowArgs;
typedef FlowArgsT<double> FlowArgsD;

template <class A> __device__ __forceinline__ bool gate_closed(const A& a) {
  return a.gate && ((*a.gate == a.gate_epoch) ? 1 : 0) != a.gate_want;
}

// boundary_slopes='circular' (RQSFlow(periodized=True), flows.py:131): the last knot's unnormalized slope IS the
// first knot's.  Run after prepare_kernel on the same stream: in the weight snapshot, column 3K of every output
// layer becomes a copy of column 2K, so the kernels produce theta[3K] == theta[2K] bit for bit with no code of
// their own (the caller's parameters are not touched).  One block per conditioner.
__global__ void circular_slopes_kernel(const float* __restrict__ params, float* __restrict__ prep, int K, int H,
                                       int D, int L, int M, int64_t per_layer) {
  const int P = 3 * K + 1;
  const int hdr = hdr_floats(K);
  const int l = blockIdx.x / (D - 1), d = 1 + blockIdx.x % (D - 1);
  int64_t o = (int64_t)l * per_layer;
  for (int dd = 1; dd < d; ++dd) o += cond_floats_p(dd, H, M, P, true);
  o += cond_floats_p(d, H, M, P, true) - ((int64_t)H * P + P);       // this conditioner's output layer
  for (int r = threadIdx.x; r <= H; r += blockDim.x)                  // H weight rows + the bias row
    prep[hdr + o + (int64_t)r * P + 3 * K] = params[P + o + (int64_t)r * P + 2 * K];
}

// ---------------------------------------------------------------------------
// prepare_kernel: params (flat, caller-owned) -> prepared model buffer.
// Thread 0 normalises the `first` spline in float64; all threads snapshot the
// conditioner weights.
// ---------------------------------------------------------------------------
__global__ void prepare_kernel(const float* __restrict__ params, float* __restrict__ prep,
                               int K, int64_t n_params, double lo, double hi, double min_bin,
                               double min_slope, int D, int L, int M, int64_t per_layer,
                               int64_t per_layer_q, int64_t mfma_off, int64_t tabd_off, int H, int periodic) {
  const int P = 3 * K + 1;
  const int hdr = hdr_floats(K);
  constexpr int MAXK = 64;
  // The LAST block normalises the `first` spline (float64); the others snapshot the weights.  The ~40 float64
  // exp / log calls of the normalisation used to run one after the other on thread 0 of block 0, behind its share of
  // the copies: 20 us for a kernel every loss evaluation starts with -- now one call per lane, and only the sums
  // (whose order fixes the result's bits) stay serial.
  if (blockIdx.x != gridDim.x - 1) {
    const int nb = gridDim.x - 1;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_params - P; i += (int64_t)nb * blockDim.x)
      prep[hdr + i] = params[P + i];
    if (mfma_off > 0) {
      // MFMA-layout copy of every conditioner (H = P = 16): see conditioner_mfma.
      const int nc = L * (D - 1);
      for (int ci = blockIdx.x; ci < nc; ci += nb) {
        const int l = ci / (D - 1), d = 1 + ci % (D - 1);
        int64_t so = P + l * per_layer, qo = mfma_off + l * per_layer_q;
        for (int dd = 1; dd < d; ++dd) { so += cond_floats(dd, 16, M, 16); qo += cond_floats_mfma(dd, M); }
        const float* src = params + so;
        float* dst = prep + qo;
        const int nf = (1 + d) + 1 + 2 * M;
        for (int idx = threadIdx.x; idx < nf * 64; idx += blockDim.x) {
          const int f = idx >> 6, lane = idx & 63, g = lane >> 4, i = lane & 15;
          float o[4];
          if (f <= d) {                         // W0 row f (f = 0: the c row)
            for (int t = 0; t < 4; ++t) o[t] = src[f * 16 + 4 * g + t];
          } else if (f == d + 1) {              // b0
            for (int t = 0; t < 4; ++t) o[t] = src[(1 + d) * 16 + 4 * g + t];
          } else {
            const int m = (f - (d + 2)) >> 1;
            const float* Wm = src + (1 + d) * 16 + 16 + m * (256 + 16);
            if (((f - (d + 2)) & 1) == 0) { for (int t = 0; t < 4; ++t) o[t] = Wm[(4 * g + t) * 16 + i]; }   // A step t
            else { for (int t = 0; t < 4; ++t) o[t] = Wm[256 + 4 * g + t]; }                               // bias rows 4g+r
          }
          for (int t = 0; t < 4; ++t) dst[(f * 64 + lane) * 4 + t] = o[t];
        }
      }
    }
    return;
  }

  __shared__ double ex[2][MAXK], dl[MAXK + 1], mxs[2];
  if (threadIdx.x < 2) {
    const float* u = params + threadIdx.x * K;
    double mx = u[0];
    for (int k = 1; k < K; ++k) mx = fmax(mx, (double)u[k]);
    mxs[threadIdx.x] = mx;
  }
  __syncthreads();
  const double offset = log(exp(1.0 - min_slope) - 1.0);
  for (int idx = threadIdx.x; idx < 3 * K + 1; idx += blockDim.x) {
    if (idx < 2 * K) {
      const int part = idx / K, k = idx - part * K;
      ex[part][k] = exp((double)params[part * K + k] - mxs[part]);
    } else {
      const int k = idx - 2 * K;
      const double v = (double)params[2 * K + (periodic && k == K ? 0 : k)] + offset;      // circular: slope K := slope 0
      dl[k] = fmax(v, 0.0) + log1p(exp(-fabs(v))) + min_slope;
    }
  }
  __syncthreads();
  __shared__ double xk[MAXK + 1], yk[MAXK + 1];
  double* td = reinterpret_cast<double*>(prep + tabd_off);     // the same table in float64
  for (int i = threadIdx.x; i < hdr; i += blockDim.x) { prep[i] = 0.0f; td[i] = 0.0; }
  const double total = (hi - lo) - K * min_bin;
  if (threadIdx.x < 2) {       // the knot positions: running sums, in the one order that fixes their bits
    const int part = threadIdx.x;
    double* pos = part == 0 ? xk : yk;
    double sum = 0;
    for (int k = 0; k < K; ++k) sum += ex[part][k];
    double run = 0;
    pos[0] = lo;
    for (int k = 0; k < K - 1; ++k) {
      run += ex[part][k] / sum * total + min_bin;
      pos[k + 1] = lo + run;
    }
    pos[K] = hi;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += blockDim.x) {      // one bin per thread
    const double bw = xk[k + 1] - xk[k], bh = yk[k + 1] - yk[k], s = bh / bw;
    td[tab_off(F_X0, K) + k] = xk[k];            td[tab_off(F_Y0, K) + k] = yk[k];
    td[tab_off(F_BW, K) + k] = bw;               td[tab_off(F_BH, K) + k] = bh;
    td[tab_off(F_IBW, K) + k] = 1.0 / bw;        td[tab_off(F_IBH, K) + k] = 1.0 / bh;
    td[tab_off(F_S, K) + k] = s;                 td[tab_off(F_ST, K) + k] = dl[k + 1] + dl[k] - 2.0 * s;
    td[tab_off(F_D0, K) + k] = dl[k];            td[tab_off(F_D1, K) + k] = dl[k + 1];
    td[tab_off(F_L2S, K) + k] = 2.0 * log(s);
    prep[tab_off(F_X0, K) + k] = (float)xk[k];
    prep[tab_off(F_Y0, K) + k] = (float)yk[k];
    prep[tab_off(F_BW, K) + k] = (float)bw;
    prep[tab_off(F_BH, K) + k] = (float)bh;
    prep[tab_off(F_IBW, K) + k] = (float)(1.0 / bw);
    prep[tab_off(F_IBH, K) + k] = (float)(1.0 / bh);
    prep[tab_off(F_S, K) + k] = (float)s;
    prep[tab_off(F_ST, K) + k] = (float)(dl[k + 1] + dl[k] - 2.0 * s);
    prep[tab_off(F_D0, K) + k] = (float)dl[k];
    prep[tab_off(F_D1, K) + k] = (float)dl[k + 1];
    prep[tab_off(F_L2S, K) + k] = (float)(2.0 * log(s));
  }
  for (int k = threadIdx.x; k <= K; k += blockDim.x) {
    const float big = 1.152921504606846976e18f;      // 2^60
    if (2 * k + 1 < 2 * tab_stride(K)) {
      prep[tab_off(F_XKB, K) + 2 * k] = prep[tab_off(F_XKB, K) + 2 * k + 1] = -(float)xk[k] * big;
      prep[tab_off(F_YKB, K) + 2 * k] = prep[tab_off(F_YKB, K) + 2 * k + 1] = -(float)yk[k] * big;
    }
    prep[tab_off(F_XK, K) + k] = (float)xk[k];
    prep[tab_off(F_YK, K) + k] = (float)yk[k];
    td[tab_off(F_XK, K) + k] = xk[k];
    td[tab_off(F_YK, K) + k] = yk[k];
  }
  if (threadIdx.x != 64) return;       // the linear tails (a lane of another wave than the bins')
  double* tld = td + tab_off(F_TAIL, K);
  tld[T_DLO] = dl[0];            tld[T_DHI] = dl[K];
  tld[T_LOG_DLO] = log(dl[0]);   tld[T_LOG_DHI] = log(dl[K]);
  tld[T_INV_DLO] = 1.0 / dl[0];  tld[T_INV_DHI] = 1.0 / dl[K];
  float* tl = prep + tab_off(F_TAIL, K);
  tl[T_DLO] = (float)dl[0];             tl[T_DHI] = (float)dl[K];
  tl[T_LOG_DLO] = (float)log(dl[0]);    tl[T_LOG_DHI] = (float)log(dl[K]);
  tl[T_INV_DLO] = (float)(1.0 / dl[0]); tl[T_INV_DHI] = (float)(1.0 / dl[K]);
}

// ---------------------------------------------------------------------------
// LDS tile: [hdr table][U: D x TS][O: D x TS], TS = 256 * SPL samples per
// workgroup.  Lane t owns samples SPL*t .. SPL*t+SPL-1 of the tile: its column
// is U[d*TS + SPL*t] (one ds_read_b32 / ds_read_b64 per dimension,
// conflict-free).
// ---------------------------------------------------------------------------
// e / D for e < 2^16 (a tile has at most 512 * 64 elements): one v_mul_hi_u32
// instead of the ~20-instruction 32-bit division sequence.
template <class R>
__device__ __forceinline__ void tile_load(const R* __restrict__ g, R* U, int D, uint32_t magic, int TS,
                                          int64_t tile_start, int64_t B) {
  const int64_t base = tile_start * D;
  const int n_el = (int)(B - tile_start < TS ? B - tile_start : TS) * D;
  for (int e = threadIdx.x; e < TS * D; e += TILE) {
    const int s = magic ? (int)__umulhi((uint32_t)e, magic) : e, d = e - s * D;   // magic 0: D = 1
    U[d * TS + s] = e < n_el ? g[base + e] : (R)0;
  }
}

// One tile of base noise straight into LDS: the tile's TS*D stream elements are
// contiguous; a thread draws whole Philox blocks (4 normals) and scatters them.
__device__ __forceinline__ void tile_noise(uint64_t seed, uint64_t first_element, float* U, int D, uint32_t magic,
                                           int TS, int64_t n_valid_samples, int nthreads = TILE) {
  const int n_el = (int)(n_valid_samples < TS ? n_valid_samples : TS) * D;
  const uint64_t blk0 = first_element >> 2;
  const int n_blk = (int)(((first_element + (uint64_t)(TS * D) - 1) >> 2) - blk0) + 1;
  for (int q = threadIdx.x; q < n_blk; q += nthreads) {
    const uint64_t blk = blk0 + (uint64_t)q;
    float z[4];
    philox_normals4(seed, blk, z);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t e = (int64_t)((blk << 2) + r) - (int64_t)first_element;
      if (e >= 0 && e < TS * D) {
        const int s = magic ? (int)__umulhi((uint32_t)e, magic) : (int)e, d = (int)e - s * D;
        U[d * TS + s] = e < n_el ? z[r] : 0.0f;
      }
    }
  }
}

__device__ __forceinline__ float normal_at(uint64_t seed, uint64_t e) {     // element e of the cnf_fill_normal stream
  float z[4];
  philox_normals4(seed, e >> 2, z);
  return z[e & 3];
}

// The tile of a seeded flow call (FlowArgsT::in == null): one contiguous run of the stream where the tile lies in
// one slice (or the slices follow each other in the stream), element by element otherwise.
__device__ __forceinline__ void tile_noise_flow(const FlowArgsT<float>& a, float* U, int D, uint32_t magic, int TS,
                                                int64_t tile_start, int nthreads) {
  const int64_t left = a.B - tile_start;
  const int64_t last = tile_start + (left < TS ? left : TS) - 1;
  const bool one = a.c_block >= a.B;
  const int64_t s0 = one ? 0 : tile_start / a.c_block, s1 = one ? 0 : last / a.c_block;
  if (s0 == s1 || a.slice_stride == a.c_block) {
    const int64_t st = a.first_sample + s0 * a.slice_stride + (tile_start - s0 * a.c_block);
    tile_noise(a.seed, (uint64_t)st * (uint64_t)D, U, D, magic, TS, left, nthreads);
    return;
  }
  const int n_el = (int)(left < TS ? left : TS) * D;
  for (int e = threadIdx.x; e < TS * D; e += nthreads) {
    const int s = magic ? (int)__umulhi((uint32_t)e, magic) : e, d = e - s * D;
    float v = 0.0f;
    if (e < n_el) {
      const int64_t i = tile_start + s, sl = i / a.c_block;
      const int64_t st = a.first_sample + sl * a.slice_stride + (i - sl * a.c_block);
      v = normal_at(a.seed, (uint64_t)st * (uint64_t)D + (uint64_t)d);
    }
    U[d * TS + s] = v;
  }
}
__device__ __forceinline__ void tile_noise_flow(const FlowArgsT<double>&, double*, int, uint32_t, int, int64_t, int) {}

template <class R>
__device__ __forceinline__ void tile_store(R* __restrict__ g, const R* U, int D, uint32_t magic, int TS,
                                           int64_t tile_start, int64_t B) {
  const int64_t base = tile_start * D;
  const int n_el = (int)(B - tile_start < TS ? B - tile_start : TS) * D;
  for (int e = threadIdx.x; e < TS * D; e += TILE) {
    const int s = magic ? (int)__umulhi((uint32_t)e, magic) : e, d = e - s * D;
    if (e < n_el) g[base + e] = U[d * TS + s];
  }
}

template <class R>
__device__ __forceinline__ R load_cond1(const FlowArgsT<R>& a, int64_t tile_start, int64_t i) {
  if (a.fd2) {                                   // evaluation point -> its base point's condition
    if (a.c_mode == C_SINGLE) return a.c[0];
    return i < a.B ? a.c[(i / a.fd2) / a.c_block] : (R)0;
  }
  switch (a.c_mode) {
    case C_SINGLE: return a.c[0];
    case C_PER_SAMPLE: return i < a.B ? a.c[i] : (R)0;
    case C_TILE_UNIFORM: return a.c[tile_start / a.c_block];
    default: return i < a.B ? a.c[i / a.c_block] : (R)0;
  }
}
template <class T> __device__ __forceinline__ T load_cond(const FlowArgsT<typename Lanes<T>::real>& a, int64_t tile_start, int64_t i);
template <> __device__ __forceinline__ float load_cond<float>(const FlowArgs& a, int64_t ts, int64_t i) {
  return load_cond1(a, ts, i);
}
template <> __device__ __forceinline__ double load_cond<double>(const FlowArgsD& a, int64_t ts, int64_t i) {
  return load_cond1(a, ts, i);
}
template <> __device__ __forceinline__ v2f load_cond<v2f>(const FlowArgs& a, int64_t ts, int64_t i) {
  if (a.c_mode == C_SINGLE || a.c_mode == C_TILE_UNIFORM) return splat<v2f>(load_cond1(a, ts, i));
  return v2f{load_cond1(a, ts, i), load_cond1(a, ts, i + 1)};
}

// finite-difference mode: the tile's evaluation points are built from the base points on the fly
template <class R>
__device__ __forceinline__ void tile_load_fd(const FlowArgsT<R>& a, R* U, int D, uint32_t magic, int TS, int64_t tile_start) {
  const int n_el = (int)(a.B - tile_start < TS ? a.B - tile_start : TS) * D;
  for (int e = threadIdx.x; e < TS * D; e += TILE) {
    const int s = magic ? (int)__umulhi((uint32_t)e, magic) : e, d = e - s * D;
    R v = (R)0;
    if (e < n_el) {
      const int64_t j = tile_start + s, i = j / a.fd2;
      const int k = (int)(j - i * a.fd2);
      v = a.in[i * D + d];
      if ((k >> 1) == d) v += (k & 1) ? -a.fd_h : a.fd_h;
    }
    U[d * TS + s] = v;
  }
}
// (log_prob(+) - log_prob(-)) / dx of the pair (j, j + 1), j even
__device__ __forceinline__ void store_fd(const FlowArgs& a, int64_t j, v2f lp) {
  if (j + 1 < a.B) a.aux[j >> 1] = (lp.x - lp.y) * a.fd_inv_dx;
}
__device__ __forceinline__ void store_fd(const FlowArgs& a, int64_t j, float lp) {
  const float other = __shfl_xor(lp, 1, 64);
  if (!(j & 1) && j + 1 < a.B) a.aux[j >> 1] = (lp - other) * a.fd_inv_dx;
}
__device__ __forceinline__ void store_fd(const FlowArgsD&, int64_t, double) {}

__device__ __forceinline__ float hsum(float v) { return v; }
__device__ __forceinline__ v2f hsum(v2f v) { return v; }

// One pass of the whole flow over the thread's own sample(s), in place in LDS.
// TO_BASE=false: base -> data (chain.inverse, spline inverse, conditions on the
// layer input: conditional.py:169-177, autoregressive.py:109-136).
// TO_BASE=true : data -> base (chain.forward, spline forward, conditions on
// already-produced outputs: conditional.py:159-167, autoregressive.py:76-107).
// Returns the accumulated log|det J|; the result is left in `U` (swapped).
// PRECISE (TO_BASE only): the precise position path of cnf_device.h; `e2tab` is its 2^(-i/32) table in LDS and
// `bacc` receives sum_d x_d^2 of the recovered base point in float64.
// DFIX > 0: the event dimension is this compile-time constant (the single-batch kernel at dim 2: the dimension
// loop, the conditioner offsets and the tile transposes lose their runtime arithmetic)
template <int H, int K, bool TO_BASE, bool FAST, class T, bool MFMA = false, bool PRECISE = false, bool PERIODIC = false, int DFIX = 0>
__device__ __forceinline__ T flow_pass(const ModelArgs& a, const typename Lanes<T>::real* tab,
                                       typename Lanes<T>::real*& U, typename Lanes<T>::real*& O, T c,
                                       const double* e2tab = nullptr, const double* tabd = nullptr,
                                       typename Lanes<T>::real* LO = nullptr, BaseAcc<T>* bacc = nullptr) {
  typedef typename Lanes<T>::real R;
  static_assert(!PRECISE || (TO_BASE && !std::is_same<T, double>::value), "precise path: data -> base, fp32 kernels");
  const SplineConstsT<R>& sc = sc_of<R>(a);
  static_assert(!MFMA || (H == 16 && K == 5), "the MFMA conditioner is built for H = 16, P = 16");
  static_assert(!PERIODIC || (!MFMA && !PRECISE), "periodized: the scalar-weight conditioner, plain positions");
  constexpr int P = 3 * K + 1;
  constexpr bool INV = !TO_BASE;
  constexpr int SPL = Lanes<T>::N;
  constexpr int TS = TILE * SPL;
  uniform_ptr weights = as_uniform(a.prep + hdr_floats(K));
  const int D = DFIX ? DFIX : a.D;
  T acc = splat<T>(0.0f);
  for (int step = 0; step < a.L; ++step) {
    const int l = TO_BASE ? a.L - 1 - step : step;
    const bool odd = l & 1;                       // flows.py:141-143 perms
    const int first_idx = odd ? D - 1 : 0, idx_step = odd ? -1 : 1;
    R* cu = U + SPL * threadIdx.x;
    R* co = O + SPL * threadIdx.x;
    T o, ld, olo;
    const bool last = step == a.L - 1;
    [[maybe_unused]] R* clo = nullptr;
    if constexpr (PRECISE) {
      // LO[d]: what rounding dimension d's value to fp32 dropped (this thread's column; in place: read as the
      // layer's input, overwritten with its output).  The data themselves are exact fp32: zero before layer 1.
      clo = LO + SPL * threadIdx.x;
      const T vlo = step == 0 ? splat<T>(0.0f) : lds_get<T>(clo, first_idx, TS);
      table_spline_precise<K, FAST>(tab, tabd, lds_get<T>(cu, first_idx, TS), vlo, sc, o, ld, olo);
      lds_put(clo, first_idx, TS, olo);
      if (last) bacc->add(o, olo);
    } else {
      table_spline<K, INV, FAST, T>(tab, lds_get<T>(cu, first_idx, TS), sc, o, ld);
    }
    lds_put(co, first_idx, TS, o);
    acc += ld;
    uniform_ptr w = weights + l * a.per_layer;
    const float* wq = a.wq + l * a.per_layer_q;
    for (int d = 1; d < D; ++d) {
      const int i = first_idx + d * idx_step;
      T th[P];
      if constexpr (MFMA && !std::is_same<T, double>::value) {
        // (at dim 2 the MFMA-layout first layer is three back-to-back vector loads: faster than scalar loads +
        // a transpose for a lone wave -- 6.1 vs 6.4 us per 65 536-sample call; from dim 3 the row loop dominates)
        conditioner_mfma<T>(reinterpret_cast<const f4*>(wq), d, a.M, c, TO_BASE ? co : cu, first_idx, idx_step, TS, th,
                            D >= 3 ? w : nullptr);
        wq += cond_floats_mfma(d, a.M);
        w += cond_floats(d, H, a.M, P);
      } else {
        conditioner<H, P, T, PERIODIC>(w, d, a.M, c, TO_BASE ? co : cu, first_idx, idx_step, TS, th);
        w += cond_floats_p(d, H, a.M, P, PERIODIC);
      }
      if constexpr (PRECISE) {
        const T vlo = step == 0 ? splat<T>(0.0f) : lds_get<T>(clo, i, TS);
        cond_spline_precise<K, FAST, false>(th, lds_get<T>(cu, i, TS), vlo, sc, a.scd, e2tab, o, ld, olo);
        lds_put(clo, i, TS, olo);
        if (last) bacc->add(o, olo);
      } else {
        cond_spline<K, INV, FAST, T>(th, lds_get<T>(cu, i, TS), sc, o, ld);
      }
      lds_put(co, i, TS, o);
      acc += ld;
    }
    R* t = U; U = O; O = t;
  }
  return acc;
}

template <class T>
__device__ __forceinline__ T base_logprob(const typename Lanes<T>::real* col, int D, int TS) {
  typedef typename Lanes<T>::real R;
  T b = splat<T>(0.0f);
  for (int d = 0; d < D; ++d) { const T x = lds_get<T>(col, d, TS); b = vfma(x * (R)-0.5, x, b); }
  return b - (R)(D * HALF_LOG_2PI);
}

__device__ __forceinline__ void store_aux(float* aux, int64_t i, int64_t B, float r) {
  if (i < B) aux[i] = r;
}
__device__ __forceinline__ void store_aux(double* aux, int64_t i, int64_t B, double r) {
  if (i < B) aux[i] = r;
}
__device__ __forceinline__ void store_aux(float* aux, int64_t i, int64_t B, v2f r) {
  if (i + 1 < B && ((reinterpret_cast<uintptr_t>(aux + i) & 7) == 0)) *reinterpret_cast<v2f*>(aux + i) = r;
  else { if (i < B) aux[i] = r.x; if (i + 1 < B) aux[i + 1] = r.y; }
}

template <int H, int K, bool TO_BASE, bool FAST, class T, bool MFMA = false, bool PRECISE = false, bool PERIODIC = false, int DFIX = 0>
__global__ __launch_bounds__(TILE, 2) void flow_kernel(const FlowArgsT<typename Lanes<T>::real> a) {
  typedef typename Lanes<T>::real R;
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  R* lds = reinterpret_cast<R*>(lds_raw);
  constexpr int HDR = hdr_floats(K);
  constexpr int SPL = Lanes<T>::N;
  constexpr int TS = TILE * SPL;
  const int DD = DFIX ? DFIX : a.m.D;
  const uint32_t dmagic = DFIX == 2 ? 0x80000000u : (uint32_t)a.div_magic;
  R* tab = lds;
  R* U = lds + HDR;
  R* O = U + DD * TS;
  if (gate_closed(a)) return;
  for (int i = threadIdx.x; i < HDR; i += TILE) tab[i] = table_of<R>(a.m)[i];
  [[maybe_unused]] double* e2tab = nullptr;
  [[maybe_unused]] double* tabd = nullptr;
  [[maybe_unused]] R* LO = nullptr;
  if constexpr (PRECISE) {      // [.. U O][LO][2^(-i/32) table][float64 `first` table]; HDR, D * TS even: 8-byte aligned
    LO = O + DD * TS;
    e2tab = reinterpret_cast<double*>(LO + DD * TS);
    tabd = e2tab + EXP2_N;
    for (int i = threadIdx.x; i < EXP2_N; i += TILE) e2tab[i] = a.m.e2tab[i];
    for (int i = threadIdx.x; i < HDR; i += TILE) tabd[i] = a.m.tabd[i];
  }

  const int64_t n_tiles = (a.B + TS - 1) / TS;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t tile_start = tile * TS;
    const int64_t i = tile_start + SPL * threadIdx.x;
    __syncthreads();                       // previous tile's stores are done with U/O
    if (TO_BASE && !PRECISE && a.fd2) tile_load_fd<R>(a, U, DD, dmagic, TS, tile_start);
    else if (a.in) tile_load<R>(a.in, U, DD, dmagic, TS, tile_start, a.B);
    else tile_noise_flow(a, U, DD, dmagic, TS, tile_start, TILE);
    const T c = load_cond<T>(a, tile_start, i);
    __syncthreads();

    T base = splat<T>(0.0f);
    if (!TO_BASE && a.aux_mode == AUX_LOGPROB && a.aux) base = base_logprob<T>(U + SPL * threadIdx.x, DD, TS);
    BaseAcc<T> bacc;
    // data -> base: the splines' clamps and bin searches turn a NaN coordinate into a finite point (log_prob(NaN) came
    // out as -58.9); the reference's arithmetic propagates it.  v - v is 0 for a finite v and NaN otherwise.
    [[maybe_unused]] T poison = splat<T>(0.0f);
    if constexpr (TO_BASE) {
      for (int d = 0; d < DD; ++d) { const T v = lds_get<T>(U + SPL * threadIdx.x, d, TS); poison += v - v; }
    }
    const T acc = flow_pass<H, K, TO_BASE, FAST, T, MFMA, PRECISE, PERIODIC, DFIX>(a.m, tab, U, O, c, e2tab, tabd, LO, &bacc);
    if (a.aux) {
      T r = acc;
      if (a.aux_mode == AUX_LOGPROB) {
        // log_prob = base(x) + ildj (conditional.py:316-321); lp_y = lp_x - fldj (:399-401)
        if constexpr (PRECISE) r = bacc.log_prob(acc, DD);
        else r = TO_BASE ? base_logprob<T>(U + SPL * threadIdx.x, DD, TS) + acc : base - acc;
      }
      if constexpr (TO_BASE) r += poison;
      if (TO_BASE && !PRECISE && a.fd2) store_fd(a, i, r);
      else store_aux(a.aux, i, a.B, r);
    }
    if (a.out) {
      if constexpr (TO_BASE) {
        for (int d = 0; d < DD; ++d) lds_put(U + SPL * threadIdx.x, d, TS, lds_get<T>(U + SPL * threadIdx.x, d, TS) + poison);
      }
      __syncthreads();
      tile_store<R>(a.out, U, DD, dmagic, TS, tile_start, a.B);
    }
  }
}


// ---------------------------------------------------------------------------
// flow_dpar_kernel: base -> data (sample / sample_and_log_prob / forward) for D >= 3 with the D - 1 conditioners
// of a layer on different WAVES.  In this direction every conditioner of a layer sees the layer's INPUTS
// (autoregressive.py:109-136: `y[perm[:d]]` of the incoming event), so they are independent of each other; the
// one-sample-per-lane kernel above runs them back to back on one lane and needs ~2 M samples to fill the chip
// (config 4's per-GPU shard is 32 768: 512 waves on 1 024 SIMDs, each with a chain of 2 x 9 conditioners).
// Here a workgroup owns a tile of 64 * SPL samples and has one wave per conditioned dimension (d = 1 + wave,
// strided if D - 1 exceeds the wave count): the weights stay wave-uniform (scalar operands), a layer costs one
// conditioner + one spline per wave and a barrier, and the shard becomes 256-512 workgroups of D - 1 waves.  The
// waves' log-det shares are summed in a fixed order (deterministic).
// ---------------------------------------------------------------------------
template <class R>
__device__ __forceinline__ void tile_load_n(const R* __restrict__ g, R* U, int D, uint32_t magic, int TS,
                                            int64_t tile_start, int64_t B, int nthreads) {
  const int64_t base = tile_start * D;
  const int n_el = (int)(B - tile_start < TS ? B - tile_start : TS) * D;
  for (int e = threadIdx.x; e < TS * D; e += nthreads) {
    const int s = magic ? (int)__umulhi((uint32_t)e, magic) : e, d = e - s * D;
    U[d * TS + s] = e < n_el ? g[base + e] : (R)0;
  }
}
template <class R>
__device__ __forceinline__ void tile_store_n(R* __restrict__ g, const R* U, int D, uint32_t magic, int TS,
                                             int64_t tile_start, int64_t B, int nthreads) {
  const int64_t base = tile_start * D;
  const int n_el = (int)(B - tile_start < TS ? B - tile_start : TS) * D;
  for (int e = threadIdx.x; e < n_el; e += nthreads) {
    const int s = magic ? (int)__umulhi((uint32_t)e, magic) : e, d = e - s * D;
    g[base + e] = U[d * TS + s];
  }
}

template <int H, int K, bool FAST, class T>
__global__ __launch_bounds__(1024) void flow_dpar_kernel(const FlowArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int HDR = hdr_floats(K), P = 3 * K + 1, SPL = Lanes<T>::N, TS = 64 * SPL;
  const int D = a.m.D, L = a.m.L, NT = blockDim.x, NW = NT >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float* tab = lds;
  float* U = lds + HDR;
  float* O = U + D * TS;
  float* LD = O + D * TS;                                  // [NW][TS] log-det shares
  if (gate_closed(a)) return;
  for (int i = threadIdx.x; i < HDR; i += NT) tab[i] = a.m.prep[i];
  const SplineConsts& sc = a.m.sc;
  uniform_ptr weights = as_uniform(a.m.prep + HDR);
  // floats of the conditioners 1 .. d-1 of a layer: sum_{q<d} ((1 + q) H + C), C = the d-independent part
  const int64_t cpart = cond_floats(0, H, a.m.M, P) - H;

  const int64_t n_tiles = (a.B + TS - 1) / TS;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t tile_start = tile * TS;
    const int64_t i = tile_start + SPL * lane;
    __syncthreads();
    if (a.in) tile_load_n<float>(a.in, U, D, a.div_magic, TS, tile_start, a.B, NT);
    else tile_noise_flow(a, U, D, a.div_magic, TS, tile_start, NT);
    const T c = load_cond<T>(a, tile_start, i);
    __syncthreads();
    T base = splat<T>(0.0f);
    if (wave == 0 && a.aux_mode == AUX_LOGPROB && a.aux) base = base_logprob<T>(U + SPL * lane, D, TS);
    T acc = splat<T>(0.0f);
    for (int l = 0; l < L; ++l) {
      const bool odd = l & 1;                              // flows.py:141-143 perms
      const int first_idx = odd ? D - 1 : 0, idx_step = odd ? -1 : 1;
      const float* cu = U + SPL * lane;
      float* co = O + SPL * lane;
      T o, ld;
      if (wave == 0) {
        table_spline<K, true, FAST, T>(tab, lds_get<T>(cu, first_idx, TS), sc, o, ld);
        lds_put(co, first_idx, TS, o);
        acc += ld;
      }
      for (int d = 1 + wave; d < D; d += NW) {
        const int64_t off = l * a.m.per_layer + (int64_t)H * ((d - 1) + (int64_t)d * (d - 1) / 2) + (d - 1) * cpart;
        T th[P];
        conditioner<H, P, T>(weights + off, d, a.m.M, c, cu, first_idx, idx_step, TS, th);
        const int idx = first_idx + d * idx_step;
        cond_spline<K, true, FAST, T>(th, lds_get<T>(cu, idx, TS), sc, o, ld);
        lds_put(co, idx, TS, o);
        acc += ld;
      }
      __syncthreads();
      float* t = U; U = O; O = t;
    }
    if (a.aux) lds_put(LD + wave * TS + SPL * lane, 0, TS, acc);
    __syncthreads();
    if (a.aux && wave == 0) {
      T tot = splat<T>(0.0f);
      for (int w = 0; w < NW; ++w) tot += lds_get<T>(LD + w * TS + SPL * lane, 0, TS);
      store_aux(a.aux, i, a.B, a.aux_mode == AUX_LOGPROB ? base - tot : tot);
    }
    if (a.out) tile_store_n<float>(a.out, U, D, a.div_magic, TS, tile_start, a.B, NT);
  }
}

// ---------------------------------------------------------------------------
// flow_pwl_kernel: the dim-2 flow with the conditioner read from the exact
// piecewise-linear tables of cnf_pwl.h (condition uniform per slice).  One
// 1024-thread workgroup per CU keeps the L tables of its current slice in LDS
// (every row when L <= 3: L x 48 KB; else L x 22 KB: header arrays + the first PWL_LROWS rows); each lane owns
// two consecutive samples, whose 4 input floats are one 16-byte load and whose
// outputs are one 16-byte + one 8-byte store -- no LDS staging of the points.
// ---------------------------------------------------------------------------
constexpr int PWL_MAX_THREADS = 1024;

struct PwlArgs {
  ModelArgs m;
  const float* in;
  float* out;
  float* aux;
  const float* tables;
  int64_t B, slice_len;
  int32_t n_slices, tiles_per_slice, aux_mode;
  const uint32_t* gate;      // see FlowArgsT
  uint32_t gate_epoch;
  int32_t gate_want;
  // in == null: base noise drawn in the kernel -- sample j of slice s (of this launch) is stream sample
  // first_sample + s * slice_stride + j of the cnf_fill_normal stream of `seed`
  uint64_t seed;
  int64_t first_sample, slice_stride;
  int32_t in_shared;         // the slices share ONE set of input points in[slice_len, 2] (the same base draw pushed to
};                           // several times: cnf_kinetic_potential_vjp)

// Every reference call site hands the flow one time broadcast to cond[B,1] (applications.py:153,226,231):
// per-sample in form, uniform in content.  For launches large enough for the table path this kernel checks it on
// the device: a block that finds c[i] != c[0] stamps the call's epoch into *flag.  The table kernels and the MLP
// kernel of the call are both enqueued and read the stamp: exactly one of them does the work (no host round trip).
__global__ void cond_uniform_kernel(const float* __restrict__ c, int64_t B, uint32_t* flag, uint32_t epoch) {
  const uint32_t c0 = __float_as_uint(c[0]);
  bool diff = false;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x)
    diff |= __float_as_uint(c[i]) != c0;
  if (__syncthreads_or(diff) && threadIdx.x == 0) *flag = epoch;
}

// The dim-2 flow on one sample pair held in registers, conditioner from the tables (`tbl`: the L
// tables in LDS, `gtbl`: the same in global memory for rows past the LDS window).  In place;
// returns the accumulated log|det J| of the direction.
// SHIFT_FREE_OK: use the shift-free spline evaluation where the sample's grid cell allows it (the flow kernels;
// the loss kernel, at its register limit with three table sets, always evaluates the general form).
// LFIX > 0: the number of flow layers is this compile-time constant -- the layer loop is unrolled, the layer's
// parity (which coordinate is conditioned on which) and its table's LDS offset are literals instead of per-layer
// selects and address arithmetic.
template <int K, bool TO_BASE, bool FAST, bool PRECISE = false, bool SHIFT_FREE_OK = false, int LROWS = PWL_LROWS, int LFIX = 0>
__device__ __forceinline__ v2f flow2_tables(const float* tab, const float* tbl, const float* __restrict__ gtbl,
                                            int L, const SplineConsts sc, v2f& u0, v2f& u1,
                                            const PreciseConsts* pc = nullptr, const double* e2tab = nullptr,
                                            const double* tabd = nullptr, BaseAcc<v2f>* bacc = nullptr) {
  constexpr bool INV = !TO_BASE;
  static_assert(!PRECISE || TO_BASE, "precise path: data -> base");
  v2f acc = splat<v2f>(0.0f);
  [[maybe_unused]] v2f lo0 = splat<v2f>(0.0f), lo1 = lo0;      // precise path: what rounding u0 / u1 to fp32 dropped
  if (LFIX) L = LFIX;
#pragma unroll
  for (int step = 0; step < (LFIX ? LFIX : L); ++step) {
    const int l = TO_BASE ? L - 1 - step : step;
    const bool odd = l & 1;                     // flows.py:141-143 perms
    const v2f uf = odd ? u1 : u0, uo = odd ? u0 : u1;
    v2f of, oo, ld, olo_f, olo_o;
    bool general;
    const float* tl = tbl + l * pwl_ltbl(LROWS);
    const float* gl = gtbl + (int64_t)l * PWL_TBL;
    if constexpr (PRECISE) {
      table_spline_precise<K, FAST>(tab, tabd, uf, odd ? lo1 : lo0, sc, of, ld, olo_f);
      if (step == L - 1) bacc->add(of, olo_f);
      acc += ld;
      v2f th[PWL_P];
      pwl_eval<LROWS>(tl, gl, of, th, general);
      cond_spline_precise<K, FAST, true>(th, uo, odd ? lo0 : lo1, sc, *pc, e2tab, oo, ld, olo_o);
      if (step == L - 1) bacc->add(oo, olo_o);
      lo0 = odd ? olo_o : olo_f;
      lo1 = odd ? olo_f : olo_o;
    } else {
      // base -> data: the conditioner sees the layer's INPUT, so its table search and row reads (a chain of three
      // dependent LDS round trips) are issued first and complete under the arithmetic of the `first` spline
      PwlRows rr;
      v2f qa[K], qb[K];
      if (!TO_BASE) {
        pwl_find<LROWS>(tl, uf, rr, general);
        rr.dua = pwl_logit_pairs<LROWS>(rr.ra, gl, rr.pa, uf.x, qa);
        rr.dub = pwl_logit_pairs<LROWS>(rr.rb, gl, rr.pb, uf.y, qb);
      }
      // One logarithm per layer where every lane takes the shift-free form: both splines return the argument of
      // their log|f'| (the derivative itself) and the product goes through one v_log_f32.  The conditioned
      // factor is bounded there (slope logits in [-3, 40], bins >= 1e-4 of a range of 20: ~1e-9 .. 1e6), so the
      // product leaves the fp32 range only for a `first` spline with derivatives beyond 1e-29 .. 1e32.
      constexpr bool LOGPROD = false;     // (measured within noise: 1.477 vs 1.480 ms per launch -- off)
      v2f larg = splat<v2f>(1.0f);
      if constexpr (LOGPROD) {
        table_spline<K, INV, FAST, v2f, true>(tab, uf, sc, of, larg);
      } else {
        table_spline<K, INV, FAST, v2f>(tab, uf, sc, of, ld);
        acc += ld;
      }
      if (TO_BASE) {
        pwl_find<LROWS>(tl, of, rr, general);
        rr.dua = pwl_logit_pairs<LROWS>(rr.ra, gl, rr.pa, of.x, qa);
        rr.dub = pwl_logit_pairs<LROWS>(rr.rb, gl, rr.pb, of.y, qb);
      }
      auto slopes = [&](int ka, int kb, v2f& ta, v2f& tb) {
        ta = pwl_slope_pair<LROWS>(rr.ra, gl, rr.pa, ka, rr.dua);
        tb = pwl_slope_pair<LROWS>(rr.rb, gl, rr.pb, kb, rr.dub);
      };
      if (!SHIFT_FREE_OK || __builtin_amdgcn_ballot_w64(general) != 0)      // wave-uniform: a lane's cell is marked
      {       // marked cells (ill-conditioned pieces, far-out inputs): the general form, and its own logarithm
        cond_spline_rows<K, INV, FAST, false, false>(qa, qb, slopes, uo, sc, oo, ld);
        if constexpr (LOGPROD) { const v2f lg = Math<FAST>::log(larg); ld += INV ? -lg : lg; }
      } else {
        cond_spline_rows<K, INV, FAST, true, LOGPROD>(qa, qb, slopes, uo, sc, oo, ld);
        if constexpr (LOGPROD) { const v2f lg = Math<FAST>::log(larg * ld); ld = INV ? -lg : lg; }
      }
    }
    acc += ld;
    u0 = odd ? oo : of;
    u1 = odd ? of : oo;
  }
  return acc;
}

template <int K, bool TO_BASE, bool FAST, bool PRECISE = false, int LROWS = PWL_LROWS, int LFIX = 0, bool SEEDED = false>
__global__ __launch_bounds__(PWL_MAX_THREADS) void flow_pwl_kernel(const PwlArgs a) {
  const int PWL_THREADS = blockDim.x, PWL_TS = 2 * PWL_THREADS;
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  constexpr int HDR = (hdr_floats(K) + 3) & ~3;
  float* tab = lds_raw;
  float* tbl = lds_raw + HDR;
  const int tid = threadIdx.x;
  const int L = LFIX ? LFIX : a.m.L;
  if (gate_closed(a)) return;
  for (int i = tid; i < hdr_floats(K); i += PWL_THREADS) tab[i] = table_of<float>(a.m)[i];
  const SplineConsts sc = sc_scalars(sc_of<float>(a.m));
  [[maybe_unused]] double* e2tab = nullptr;
  [[maybe_unused]] double* tabd = nullptr;
  if constexpr (PRECISE) {                 // after the L tables (HDR and PWL_LTBL are even: 8-byte aligned)
    e2tab = reinterpret_cast<double*>(tbl + L * pwl_ltbl(LROWS));
    tabd = e2tab + EXP2_N;
    for (int i = tid; i < EXP2_N; i += PWL_THREADS) e2tab[i] = a.m.e2tab[i];
    for (int i = tid; i < hdr_floats(K); i += PWL_THREADS) tabd[i] = a.m.tabd[i];
  }

  const int total = a.n_slices * a.tiles_per_slice;
  const int per_block = (total + gridDim.x - 1) / gridDim.x;
  const int t0 = blockIdx.x * per_block;
  const int t1 = t0 + per_block < total ? t0 + per_block : total;
  int cur = -1;
  // Tile geometry is wave-uniform (scalar registers): the slice, the tile's first sample and how many of its
  // PWL_TS samples exist.  A lane's share is then a 32-bit offset from a scalar base address, and a full tile
  // -- every tile but the last of a slice of odd size -- takes the unmasked path.
  struct Tile { int slice; int valid; int64_t g0; int64_t st0; int64_t gin; };
  auto tile_of = [&](int tile) {
    Tile t;
    t.slice = tile / a.tiles_per_slice;
    const int64_t s0 = (int64_t)t.slice * a.slice_len;
    const int64_t len = a.B - s0 < a.slice_len ? a.B - s0 : a.slice_len;
    const int64_t jt = (int64_t)(tile - t.slice * a.tiles_per_slice) * PWL_TS;
    const int64_t left = len - jt;
    t.valid = left >= PWL_TS ? PWL_TS : (left > 0 ? (int)left : 0);
    t.g0 = s0 + jt;
    t.gin = a.in_shared ? jt : t.g0;
    t.st0 = a.first_sample + (int64_t)t.slice * a.slice_stride + jt;      // (seeded calls: the tile's first stream sample)
    return t;
  };
  const uint32_t lane2 = 2u * (uint32_t)tid;             // the lane's first sample within the tile
  auto tile_points = [&](const Tile& t) {
    f4 x = {0.f, 0.f, 0.f, 0.f};
    if (SEEDED && !a.in) {   // the pair's four normals (one Philox block when the pair starts on an even sample)
      const uint64_t e0 = (uint64_t)(t.st0 + lane2) * 2u;
      if ((int)lane2 < t.valid) {
        if ((e0 & 3) == 0) {
          float z[4];
          philox_normals4(a.seed, e0 >> 2, z);
#pragma unroll
          for (int q = 0; q < 4; ++q) x[q] = z[q];
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) x[q] = normal_at(a.seed, e0 + q);
        }
        if ((int)lane2 + 1 >= t.valid) { x[2] = 0.f; x[3] = 0.f; }
      }
      return x;
    }
    const float* p = a.in + 2 * t.gin;
    if (t.valid == PWL_TS) x = *reinterpret_cast<const f4*>(p + 2u * lane2);
    else if ((int)lane2 + 1 < t.valid) x = *reinterpret_cast<const f4*>(p + 2u * lane2);
    else if ((int)lane2 < t.valid) { x[0] = p[2u * lane2]; x[1] = p[2u * lane2 + 1]; }
    return x;
  };
  // The points of tile i + 1 are requested before tile i is computed: measured (r02e PMC) a wave spent half its
  // time in s_waitcnt, much of it on this one load issued right in front of its first use.
  [[maybe_unused]] f4 xn = {0.f, 0.f, 0.f, 0.f};
  if (t0 < t1) xn = tile_points(tile_of(t0));
  for (int tile = t0; tile < t1; ++tile) {
    const Tile tl = tile_of(tile);
    const int slice = tl.slice;
    if (slice != cur) {
      __syncthreads();
      pwl_stage<LROWS>(tbl, a.tables + (int64_t)slice * L * PWL_TBL, L, tid, PWL_THREADS);
      cur = slice;
      __syncthreads();
    }
    const bool full = tl.valid == PWL_TS;
    const bool v0 = (int)lane2 < tl.valid, v1 = (int)lane2 + 1 < tl.valid;
    const f4 x = xn;
    if (tile + 1 < t1) xn = tile_points(tile_of(tile + 1));
    __builtin_amdgcn_sched_barrier(0);
    v2f u0 = {x[0], x[2]}, u1 = {x[1], x[3]};
    [[maybe_unused]] v2f poison = splat<v2f>(0.0f);      // (flow_kernel: a NaN coordinate must come out as NaN)
    if constexpr (TO_BASE) poison = (u0 - u0) + (u1 - u1);

    v2f base = splat<v2f>(0.0f);
    if (!TO_BASE && a.aux_mode == AUX_LOGPROB && a.aux) base = (u0 * u0 + u1 * u1) * -0.5f - (float)(2 * HALF_LOG_2PI);
    BaseAcc<v2f> bacc;
    const v2f acc = flow2_tables<K, TO_BASE, FAST, PRECISE, true, LROWS, LFIX>(tab, tbl, a.tables + (int64_t)slice * L * PWL_TBL, L, sc,
                                                            u0, u1, &a.m.scd, e2tab, tabd, &bacc);
    if (a.aux) {
      v2f r = acc;
      if constexpr (PRECISE) { if (a.aux_mode == AUX_LOGPROB) r = bacc.log_prob(acc, 2); }
      else if (a.aux_mode == AUX_LOGPROB)
        r = TO_BASE ? (u0 * u0 + u1 * u1) * -0.5f - (float)(2 * HALF_LOG_2PI) + acc : base - acc;
      if constexpr (TO_BASE) r += poison;
      float* q = a.aux + tl.g0;
      if (full || v1) *reinterpret_cast<v2f*>(q + lane2) = r;
      else if (v0) q[lane2] = r.x;
    }
    if (a.out) {
      float* q = a.out + 2 * tl.g0;
      if constexpr (TO_BASE) { u0 += poison; u1 += poison; }
      if (full || v1) *reinterpret_cast<f4*>(q + 2u * lane2) = f4{u0.x, u1.x, u0.y, u1.y};
      else if (v0) { q[2u * lane2] = u0.x; q[2u * lane2 + 1] = u1.x; }
    }
  }
}

// ---------------------------------------------------------------------------
// loss_kernel: fused Monte-Carlo loss terms (cnf_ot/mfc/applications.py
// :129-374, cnf_ot/utils.py:311-389).  One tile of samples of one time-slice
// per workgroup iteration; several flow passes share the tile's base noise in
// LDS (the reference reuses one rng for them: applications.py:233-239); only
// one double per tile leaves the chip (atomicAdd into sums[slice]).
// LDS: [hdr][N noise][U][O][V velocity][R r3], each D x TS.
// ---------------------------------------------------------------------------
struct LossArgs {
  ModelArgs m;
  CnfLossSpec spec;
  const float* pts;     // base noise (or data points for NEG_LOGPROB)
  const float* t;       // [n_slices]
  double* sums;         // [n_slices]
  int64_t B;            // samples per slice
  int64_t n_slices;
  int64_t pts_slice_stride;   // samples between slices in pts (0: shared draw)
  uint32_t div_magic;
  // pts == nullptr: base noise is generated in the kernel (Philox stream of
  // fill_normal_kernel): sample i of slice s is stream sample first_sample + s * pts_slice_stride + i
  uint64_t seed;
  int64_t first_sample;
};

template <class T>
__device__ __forceinline__ void copy_cols(float* dst, const float* src, int D, int TS) {
  for (int d = 0; d < D; ++d) lds_put(dst, d, TS, lds_get<T>(src, d, TS));
}

template <bool FAST, class T>
__device__ __forceinline__ T potential_of(const float* y, int D, int TS, int subtype, float a) {
  using M = Math<FAST>;
  if (subtype == CNF_POT_DOUBLE_WELL) {      // (|r-a1| |r+a1| / 2)^2, applications.py:184-188
    T sm = splat<T>(0.0f), sp = splat<T>(0.0f);
    for (int d = 0; d < D; ++d) {
      const T r = lds_get<T>(y, d, TS);
      sm = vfma(r - a, r - a, sm);
      sp = vfma(r + a, r + a, sp);
    }
    return sm * sp * 0.25f;
  }
  T s2 = splat<T>(0.0f);
  for (int d = 0; d < D; ++d) { const T r = lds_get<T>(y, d, TS); s2 = vfma(r, r, s2); }
  if (subtype == CNF_POT_OBSTACLE) return M::exp(s2 * -0.5f) * 50.0f;   // applications.py:190-191
  return s2 * 0.5f;                                                      // quadratic, :181-182
}

__device__ __forceinline__ float mask_tail(float v, int64_t i, int64_t B) { return i < B ? v : 0.0f; }
__device__ __forceinline__ float mask_tail(v2f v, int64_t i, int64_t B) {
  return (i < B ? v.x : 0.0f) + (i + 1 < B ? v.y : 0.0f);
}

template <int H, int K, bool FAST, class T>
__global__ __launch_bounds__(TILE, 2) void loss_kernel(const LossArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int HDR = hdr_floats(K);
  constexpr int SPL = Lanes<T>::N;
  constexpr int TS = TILE * SPL;
  using M = Math<FAST>;
  const int D = a.m.D;
  float* tab = lds;
  float* Nn = lds + HDR;
  float* U = Nn + D * TS;
  float* O = U + D * TS;
  float* V = O + D * TS;
  float* R = V + D * TS;
  for (int i = threadIdx.x; i < HDR; i += TILE) tab[i] = a.m.prep[i];
  const int col = SPL * threadIdx.x;
  const int kind = a.spec.kind;

  const int64_t tiles_per_slice = (a.B + TS - 1) / TS;
  const int64_t n_tiles = tiles_per_slice * a.n_slices;
  SliceSum ssum;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t slice = tile / tiles_per_slice;
    const int64_t tile_start = (tile - slice * tiles_per_slice) * TS;
    const int64_t i = tile_start + col;
    __syncthreads();
    if (a.pts)
      tile_load<float>(a.pts + slice * a.pts_slice_stride * D, Nn, D, a.div_magic, TS, tile_start, a.B);
    else
      tile_noise(a.seed, (uint64_t)(a.first_sample + slice * a.pts_slice_stride + tile_start) * (uint64_t)D, Nn, D,
                 a.div_magic, TS, a.B - tile_start);
    const float t = a.t[slice];
    __syncthreads();

    // Each flow direction is instantiated ONCE (a pass is ~2.5k instructions;
    // one inlined copy per use would overflow the 64 KB instruction cache):
    // the base->data passes run in a loop over (condition, destination), the
    // data->base passes in a loop over (dimension, sign).
    T acc = splat<T>(0.0f);
    const float dt = a.spec.dt;
    const bool kin = kind <= CNF_TERM_FLOW_MATCHING;
    const int n_fwd = kind == CNF_TERM_NEG_LOGPROB ? 0 : (kind == CNF_TERM_KINETIC ? 2 : (kin ? 3 : 1));
    T fldj = splat<T>(0.0f);
    for (int p = 0; p < n_fwd; ++p) {
      const float c = !kin ? t : (p == 0 ? t - 0.5f * dt : (p == 1 ? t + 0.5f * dt : t));
      copy_cols<T>(U + col, Nn + col, D, TS);
      fldj = flow_pass<H, K, false, FAST, T>(a.m, tab, U, O, splat<T>(c));
      if (kin) {
        if (p == 0) copy_cols<T>(V + col, U + col, D, TS);                       // r1
        else if (p == 1) {                                                      // velocity = (r2 - r1)/dt
          const float inv_dt = 1.0f / dt;
          for (int d = 0; d < D; ++d)
            lds_put(V + col, d, TS, (lds_get<T>(U + col, d, TS) - lds_get<T>(V + col, d, TS)) * inv_dt);
        } else copy_cols<T>(R + col, U + col, D, TS);                            // r3
      }
    }
    if (kind == CNF_TERM_KINETIC) {
      for (int d = 0; d < D; ++d) { const T v = lds_get<T>(V + col, d, TS); acc = vfma(v, v, acc); }
    } else if (kind == CNF_TERM_POTENTIAL) {
      acc = potential_of<FAST, T>(U + col, D, TS, a.spec.subtype, a.spec.a);
    } else if (kind == CNF_TERM_REVERSE_KL) {
      const T lp = base_logprob<T>(Nn + col, D, TS) - fldj;
      T s2 = splat<T>(0.0f);
      for (int d = 0; d < D; ++d) { const T r = lds_get<T>(U + col, d, TS); s2 = vfma(r, r, s2); }
      // log(N(y;0,vs I) ws + N(y;0,vt I) wt) as a log-sum-exp (applications.py:136-163)
      const float Tt = a.spec.T, vs = 2.0f / a.spec.beta * (Tt + 1.0f), vt = 2.0f / a.spec.beta;
      const float ws = (Tt - t) / Tt, wt = t / Tt;
      const float ls = -0.5f * D * logf(6.283185307179586f * vs), lt = -0.5f * D * logf(6.283185307179586f * vt);
      const T as = vfma(s2, splat<T>(-0.5f / vs), splat<T>(ls));
      const T at = vfma(s2, splat<T>(-0.5f / vt), splat<T>(lt));
      const T mx = vmax(as, at);
      const T mix = M::exp(as - mx) * ws + M::exp(at - mx) * wt;
      acc = lp - (mx + M::log(mix));
    }
    // data->base passes: NEG_LOGPROB (one, on the points themselves) or the
    // central differences of log_prob at r3 +- dx/2 e_d (applications.py:264-273)
    const bool neg = kind == CNF_TERM_NEG_LOGPROB;
    const int n_tb = neg ? 1 : ((kind == CNF_TERM_KINETIC_SCORE || kind == CNF_TERM_FLOW_MATCHING) ? 2 * D : 0);
    const float dx = a.spec.dx;
    T lp0 = splat<T>(0.0f);
    for (int e = 0; e < n_tb; ++e) {
      const int d = e >> 1, sgn = e & 1;
      copy_cols<T>(U + col, (neg ? Nn : R) + col, D, TS);
      if (!neg) lds_put(U + col, d, TS, lds_get<T>(R + col, d, TS) + (sgn == 0 ? 0.5f * dx : -0.5f * dx));
      const T ildj = flow_pass<H, K, true, FAST, T>(a.m, tab, U, O, splat<T>(t));
      const T lp = base_logprob<T>(U + col, D, TS) + ildj;
      if (neg) acc = -lp;
      else if (sgn == 0) lp0 = lp;
      else {
        T v = vfma((lp0 - lp) * (1.0f / dx), splat<T>(a.spec.coef), lds_get<T>(V + col, d, TS));
        if (kind == CNF_TERM_FLOW_MATCHING) v -= drift_of<T>(R + col, d, D, TS, a.spec.subtype, a.spec.a);
        acc = vfma(v, v, acc);
      }
    }
    // tile reduction: lanes -> wave (shuffles) -> the wave's running sum of the slice
    float part = mask_tail(acc, i, a.B);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    ssum.add(a.sums, slice, part);
  }
  ssum.flush(a.sums);
}


// ---------------------------------------------------------------------------
// loss_pwl_kernel: the fused loss terms at dim 2 on the conditioner tables.
// Same terms and arithmetic as loss_kernel; a sample pair lives in registers,
// the passes of a term use up to three table sets (conditions t - dt/2,
// t + dt/2, t), each L x 22 KB in LDS.  Base noise comes from `pts` or from the
// Philox stream (one counter block = the pair's four normals when aligned).
// ---------------------------------------------------------------------------
struct LossPwlArgs {
  ModelArgs m;
  CnfLossSpec spec;
  const float* pts;
  const float* t;
  double* sums;
  const float* tables;        // [n_sets][n_slices][L][PWL_TBL]
  int64_t B, n_slices, pts_slice_stride;
  uint64_t seed;
  int64_t first_sample;
  int32_t n_sets, tiles_per_slice;
};

template <int K, bool FAST>
__global__ __launch_bounds__(PWL_MAX_THREADS) void loss_pwl_kernel(const LossPwlArgs a) {
  using M = Math<FAST>;
  typedef v2f T;
  const int NT = blockDim.x, TS = 2 * NT;
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  constexpr int HDR = (hdr_floats(K) + 3) & ~3;
  const int tid = threadIdx.x;
  const int L = a.m.L;
  float* tab = lds_raw;
  float* tbl = lds_raw + HDR;                               // n_sets x L tables
  float* R = tbl + a.n_sets * L * PWL_LTBL;                 // 2 x TS scratch columns (potential_of / drift_of read LDS columns)
  for (int i = tid; i < hdr_floats(K); i += NT) tab[i] = table_of<float>(a.m)[i];
  const SplineConsts sc = sc_scalars(sc_of<float>(a.m));
  const int kind = a.spec.kind;
  const bool kin = kind <= CNF_TERM_FLOW_MATCHING;
  const float dt = a.spec.dt, dx = a.spec.dx;
  const int64_t set_stride = a.n_slices * L * (int64_t)PWL_TBL;
  const int col = 2 * tid;

  const int total = (int)a.n_slices * a.tiles_per_slice;
  const int per_block = (total + gridDim.x - 1) / gridDim.x;
  const int t0 = blockIdx.x * per_block;
  const int t1 = t0 + per_block < total ? t0 + per_block : total;
  int cur = -1;
  SliceSum ssum;
  for (int tile = t0; tile < t1; ++tile) {
    const int slice = tile / a.tiles_per_slice;
    __syncthreads();                                        // the previous tile is done with R (and the tables)
    if (slice != cur) {
      for (int s = 0; s < a.n_sets; ++s)
        pwl_stage(tbl + s * L * PWL_LTBL, a.tables + s * set_stride + (int64_t)slice * L * PWL_TBL, L, tid, NT);
      cur = slice;
      __syncthreads();
    }
    const float* gslice = a.tables + (int64_t)slice * L * PWL_TBL;
    const int64_t j = (int64_t)(tile - slice * a.tiles_per_slice) * TS + col;
    const bool v0 = j < a.B, v1 = j + 1 < a.B;
    const float t = a.t[slice];
    // the pair's points: x = (n0.x, n1.x), y = (n0.y, n1.y)
    f4 x = {0.f, 0.f, 0.f, 0.f};
    const int64_t g = slice * a.pts_slice_stride + j;       // sample index in pts / offset in the stream
    if (a.pts) {
      const float* src = a.pts + 2 * g;
      if (v1 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) x = *reinterpret_cast<const f4*>(src);
      else { if (v0) { x[0] = src[0]; x[1] = src[1]; } if (v1) { x[2] = src[2]; x[3] = src[3]; } }
    } else {
      const uint64_t e0 = (uint64_t)(a.first_sample + g) * 2u;
      if ((e0 & 3) == 0) {                                  // one Philox block holds the pair
        float z[4];
        philox_normals4(a.seed, e0 >> 2, z);
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = z[q];
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = normal_at(a.seed, e0 + q);
      }
      if (!v0) { x[0] = 0.f; x[1] = 0.f; }
      if (!v1) { x[2] = 0.f; x[3] = 0.f; }
    }
    const T n0 = {x[0], x[2]}, n1 = {x[1], x[3]};

    T acc = splat<T>(0.0f);
    const int n_fwd = kind == CNF_TERM_NEG_LOGPROB ? 0 : (kind == CNF_TERM_KINETIC ? 2 : (kin ? 3 : 1));
    T fldj = splat<T>(0.0f);
    T y0 = n0, y1 = n1, va = splat<T>(0.0f), vb = splat<T>(0.0f);     // (va, vb): r1, then the velocity
    for (int p = 0; p < n_fwd; ++p) {                                  // set p: conditions t - dt/2, t + dt/2, t (kin) or t
      y0 = n0; y1 = n1;
      fldj = flow2_tables<K, false, FAST, false, true>(tab, tbl + p * L * PWL_LTBL, gslice + p * set_stride, L, sc, y0, y1);
      if (kin) {
        if (p == 0) { va = y0; vb = y1; }
        else if (p == 1) { const float inv_dt = 1.0f / dt; va = (y0 - va) * inv_dt; vb = (y1 - vb) * inv_dt; }
      }
    }
    const int tset = kin ? 2 : 0;                                       // the set of condition t
    if (kind == CNF_TERM_KINETIC) {
      acc = vfma(va, va, vb * vb);
    } else if (kind == CNF_TERM_POTENTIAL) {
      lds_put(R + col, 0, TS, y0); lds_put(R + col, 1, TS, y1);
      acc = potential_of<FAST, T>(R + col, 2, TS, a.spec.subtype, a.spec.a);
    } else if (kind == CNF_TERM_REVERSE_KL) {
      const T lp = vfma(n0 * -0.5f, n0, n1 * n1 * -0.5f) - (float)(2 * HALF_LOG_2PI) - fldj;
      const T s2 = vfma(y0, y0, y1 * y1);
      // log(N(y;0,vs I) ws + N(y;0,vt I) wt) as a log-sum-exp (applications.py:136-163)
      const float Tt = a.spec.T, vs = 2.0f / a.spec.beta * (Tt + 1.0f), vt = 2.0f / a.spec.beta;
      const float ws = (Tt - t) / Tt, wt = t / Tt;
      const float ls = -logf(6.283185307179586f * vs), lt = -logf(6.283185307179586f * vt);    // -0.5 D log(2 pi v), D = 2
      const T as = vfma(s2, splat<T>(-0.5f / vs), splat<T>(ls));
      const T at = vfma(s2, splat<T>(-0.5f / vt), splat<T>(lt));
      const T mx = vmax(as, at);
      const T mix = M::exp(as - mx) * ws + M::exp(at - mx) * wt;
      acc = lp - (mx + M::log(mix));
    }
    // data->base passes: NEG_LOGPROB (one, on the points themselves) or the central differences of
    // log_prob at r3 +- dx/2 e_d (applications.py:264-273); r3 = (y0, y1) of the pass at condition t
    const bool neg = kind == CNF_TERM_NEG_LOGPROB;
    const int n_tb = neg ? 1 : ((kind == CNF_TERM_KINETIC_SCORE || kind == CNF_TERM_FLOW_MATCHING) ? 4 : 0);
    if (kind == CNF_TERM_FLOW_MATCHING) { lds_put(R + col, 0, TS, y0); lds_put(R + col, 1, TS, y1); }
    T lp0 = splat<T>(0.0f);
    for (int e = 0; e < n_tb; ++e) {
      const int d = e >> 1, sgn = e & 1;
      T u0 = neg ? n0 : y0, u1 = neg ? n1 : y1;
      if (!neg) {
        const float h = sgn == 0 ? 0.5f * dx : -0.5f * dx;
        if (d == 0) u0 = u0 + h; else u1 = u1 + h;
      }
      const T ildj = flow2_tables<K, true, FAST, false, true>(tab, tbl + tset * L * PWL_LTBL, gslice + tset * set_stride, L, sc, u0, u1);
      const T lp = vfma(u0 * -0.5f, u0, u1 * u1 * -0.5f) - (float)(2 * HALF_LOG_2PI) + ildj;
      if (neg) acc = -lp;
      else if (sgn == 0) lp0 = lp;
      else {
        T v = vfma((lp0 - lp) * (1.0f / dx), splat<T>(a.spec.coef), d == 0 ? va : vb);
        if (kind == CNF_TERM_FLOW_MATCHING) v -= drift_of<T>(R + col, d, 2, TS, a.spec.subtype, a.spec.a);
        acc = vfma(v, v, acc);
      }
    }
    // tile reduction: lanes -> wave (shuffles) -> the wave's running sum of the slice
    float part = (v0 ? acc.x : 0.0f) + (v1 ? acc.y : 0.0f);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    ssum.add(a.sums, slice, part);
  }
  ssum.flush(a.sums);
}

// ---------------------------------------------------------------------------
// Base noise: Philox4x32-10 + Box-Muller; one counter block (4 normals) per
// thread.  Element e of the stream uses block e>>2, word pair (e&3)>>1.
// ---------------------------------------------------------------------------
// seed_dev (optional): the key is read from device memory -- state[1] of a training step's device-side state
// (cnf_step_begin) -- so that a captured step draws new noise on every replay
__global__ void fill_normal_kernel(uint64_t seed, uint64_t first_element, int64_t n,
                                   float* __restrict__ out, const uint64_t* __restrict__ seed_dev) {
  if (seed_dev) seed = seed_dev[1];
  const uint64_t first_blk = first_element >> 2;
  const uint64_t last_blk = (first_element + (uint64_t)n - 1) >> 2;
  for (uint64_t blk = first_blk + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; blk <= last_blk;
       blk += (uint64_t)gridDim.x * blockDim.x) {
    float z[4];
    philox_normals4(seed, blk, z);
    const uint64_t e0 = blk << 2;
    if (e0 >= first_element && e0 + 3 < first_element + (uint64_t)n && (((e0 - first_element) & 3) == 0) &&
        ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
      *reinterpret_cast<float4*>(out + (e0 - first_element)) = make_float4(z[0], z[1], z[2], z[3]);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint64_t e = e0 + r;
        if (e >= first_element && e < first_element + (uint64_t)n) out[e - first_element] = z[r];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// JAX-compatible base draw: jax.random.normal(key, (n, D), float64) as the reference makes it
// (conditional.py:378,399 -> distrax Normal -> jax.random.normal; float64 because solvers.py:23 enables x64), for the
// classic (non-"partitionable") threefry bit generation: element j of the flattened [size] draw takes the 64 bits
// (o0 << 32) | o1 of the Threefry-2x32-20 block with counter (j, size + j) and key (k0, k1)
// [threefry_2x32 splits the iota of 2 size counters in halves; the 64-bit combine takes the halves of the output],
// maps them to a uniform in [nextafter(-1, 0), 1) through the mantissa of a double in [1, 2), and returns
// sqrt(2) erfinv(u).  The Threefry function is pinned by the Random123 known-answer vectors (tests); the bit ->
// normal mapping restates jax._src.random (un-pinned JAX version, not installable here: cannot be compared with
// JAX itself -- "parity unpinned" for this entry point).
// ---------------------------------------------------------------------------
__host__ __device__ inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__host__ __device__ inline void threefry2x32_20(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t& o0,
                                                uint32_t& o1) {
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  const int rot[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
  for (int g = 0; g < 5; ++g) {
    for (int r = 0; r < 4; ++r) { x0 += x1; x1 = rotl32(x1, rot[g & 1][r]); x1 ^= x0; }
    x0 += ks[(g + 1) % 3];
    x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
  o0 = x0; o1 = x1;
}

__global__ __launch_bounds__(256) void fill_normal_threefry_kernel(uint32_t k0, uint32_t k1, uint64_t size, uint64_t first, int64_t n,
                                            float* __restrict__ out32, double* __restrict__ out64) {
  const double lo = -0.99999999999999988897769753748;        // nextafter(-1, 0)
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t j = first + (uint64_t)i;
    uint32_t o0, o1;
    threefry2x32_20(k0, k1, (uint32_t)j, (uint32_t)(size + j), o0, o1);
    const uint64_t bits = ((uint64_t)o0 << 32) | (uint64_t)o1;
    const double f = __longlong_as_double((long long)((bits >> 12) | 0x3FF0000000000000ull)) - 1.0;
    const double u = fmax(lo, f * (1.0 - lo) + lo);
    const double z = 1.41421356237309504880 * erfinv(u);
    if (out64) out64[i] = z;
    if (out32) out32[i] = (float)z;
  }
}

}  // namespace cnf

// ===========================================================================
// C ABI
// ===========================================================================
using namespace cnf;



static int config_valid(const CnfConfig* c) {
  if (!c) return 0;
  if (c->dim < 1 || c->dim > 64) return 0;
  if (c->num_layers < 1 || c->num_layers > 64) return 0;
  if (c->hidden_size < 1 || c->mlp_num_layers < 1 || c->mlp_num_layers > 16) return 0;
  if (c->num_bins < 1 || c->num_bins > 64) return 0;
  if (!(c->range_min < c->range_max)) return 0;
  if (!(c->min_bin_size > 0.f) || !(c->min_knot_slope > 0.f) || !(c->min_knot_slope < 1.f)) return 0;
  if (c->num_bins * c->min_bin_size > c->range_max - c->range_min) return 0;   // distrax raises
  if (c->periodized != 0 && c->periodized != 1) return 0;
  return 1;
}

extern "C" int cnf_config_supported(const CnfConfig* c) {
  if (!config_valid(c)) return 0;
#define X(HH, KK) if (c->hidden_size == HH && c->num_bins == KK) return 1;
  CNF_KERNEL_CONFIGS(X)
#undef X
  return 0;
}

extern "C" void cnf_config_default(CnfConfig* c, int32_t dim) {
  if (!c) return;
  c->dim = dim; c->num_layers = 2; c->hidden_size = 16; c->mlp_num_layers = 2; c->num_bins = 5;
  c->range_min = -10.f; c->range_max = 10.f; c->min_bin_size = 1e-4f; c->min_knot_slope = 1e-4f;
  c->periodized = 0;
}

extern "C" int64_t cnf_param_count(const CnfConfig* c) {
  if (!config_valid(c)) return CNF_ERR_INVALID;
  const int P = 3 * c->num_bins + 1;
  int64_t n = P;
  for (int d = 1; d < c->dim; ++d)
    n += (int64_t)c->num_layers * cond_floats_p(d, c->hidden_size, c->mlp_num_layers, P, c->periodized != 0);
  return n;
}

extern "C" const char* cnf_strerror(int code) {
  switch (code) {
    case CNF_OK: return "ok";
    case CNF_ERR_INVALID: return "invalid argument";
    case CNF_ERR_UNSUPPORTED: return "no kernel compiled for this (hidden_size, num_bins)";
    case CNF_ERR_NOMEM: return "out of memory";
    case CNF_ERR_HIP: return "HIP runtime error";
    default: return "unknown error";
  }
}

extern "C" const char* cnf_build_arch(void) { return "gfx950"; }

extern "C" int cnf_model_create(const CnfConfig* cfg, CnfModel** out) {
  if (!out) return CNF_ERR_INVALID;
  *out = nullptr;
  if (!config_valid(cfg)) return CNF_ERR_INVALID;
  if (!cnf_config_supported(cfg)) return CNF_ERR_UNSUPPORTED;
  CnfModel* m = new (std::nothrow) CnfModel();       // value-initialised: scalars and pointers start at zero
  if (!m) return CNF_ERR_NOMEM;
  m->cfg = *cfg;
  const int K = cfg->num_bins, P = 3 * K + 1;
  m->n_params = cnf_param_count(cfg);
  m->per_layer = 0;
  for (int d = 1; d < cfg->dim; ++d)
    m->per_layer += cond_floats_p(d, cfg->hidden_size, cfg->mlp_num_layers, P, cfg->periodized != 0);
  m->sc.lo = cfg->range_min; m->sc.hi = cfg->range_max;
  m->sc.min_bin = cfg->min_bin_size; m->sc.min_slope = cfg->min_knot_slope;
  m->sc.span_eff = (float)(((double)cfg->range_max - (double)cfg->range_min) - (double)K * (double)cfg->min_bin_size);
  m->sc.sp_offset = (float)log(exp(1.0 - (double)cfg->min_knot_slope) - 1.0);
  m->fast_math = 1;
  if (hipGetDevice(&m->device) != hipSuccess) { delete m; return CNF_ERR_HIP; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, m->device) != hipSuccess) { delete m; return CNF_ERR_HIP; }
  m->num_cus = prop.multiProcessorCount;
  // The MFMA conditioner: fp32 MFMA and fp32 VALU do not overlap on gfx950 (their busy times add up:
  // profiles/r01c), so at equal peak rate the packed-VALU conditioner is 4-5 % faster once the chip is full.  A
  // launch of one wave per SIMD is a different regime: a lone wave issues one VALU instruction per 8 cycles
  // (profiles/r01_issue_probe), and the MFMA form has ~45 % fewer of them and no scalar weight loads to wait
  // for -- 6.1 vs 9.8 us per 65 536-sample call (profiles/r02_experiments/exp_latency.log).  2 = by launch size.
  m->use_mfma = 2;
  m->use_pwl = 1;
  m->use_dpar = 1;
  // (D = 1 would need 2^32: encoded as 0, tile_load/tile_store take s = e)
  m->div_magic = cfg->dim == 1 ? 0u : (uint32_t)((((uint64_t)1 << 32) + (uint64_t)cfg->dim - 1) / (uint64_t)cfg->dim);
  m->per_layer_q = 0; m->mfma_off = 0;
  size_t q_floats = 0;
  if (cfg->hidden_size == 16 && P == 16 && cfg->dim > 1 && !cfg->periodized) {      // (no MFMA form of the sin / cos layer)
    for (int d = 1; d < cfg->dim; ++d) m->per_layer_q += cond_floats_mfma(d, cfg->mlp_num_layers);
    m->mfma_off = (hdr_floats(K) + (m->n_params - P) + 3) & ~(int64_t)3;
    q_floats = (size_t)m->per_layer_q * cfg->num_layers;
  }
  // float64 copy of the `first` table (exact-mode kernels), 8-byte aligned, after everything else
  m->tabd_off = (hdr_floats(K) + (m->n_params - P) + 4 + (int64_t)q_floats + 1) & ~(int64_t)1;
  // 2^(-i/32), i = 0 .. 1024, float64: the table of the precise position path (cnf_device.h)
  m->e2_off = m->tabd_off + 2 * hdr_floats(K);
  m->precise = 1;
  const size_t bytes = (size_t)(m->e2_off + 2 * cnf::EXP2_N) * sizeof(float) + 64;
  m->scd.lo = (double)cfg->range_min; m->scd.hi = (double)cfg->range_max;
  m->scd.min_bin = (double)cfg->min_bin_size; m->scd.min_slope = (double)cfg->min_knot_slope;
  m->scd.span_eff = (m->scd.hi - m->scd.lo) - (double)K * m->scd.min_bin;
  m->scd.sp_offset = log(exp(1.0 - m->scd.min_slope) - 1.0);
  if (hipMalloc((void**)&m->prep, bytes) != hipSuccess) { delete m; return CNF_ERR_NOMEM; }
  {
    double e2[cnf::EXP2_N];
    for (int i = 0; i < cnf::EXP2_N; ++i) e2[i] = exp2(-(double)i / (double)cnf::EXP2_STEPS);
    if (hipMemcpy(m->prep + m->e2_off, e2, sizeof(e2), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(m->prep); delete m; return CNF_ERR_HIP;
    }
  }
  if (hipEventCreateWithFlags(&m->prep_event, hipEventDisableTiming) != hipSuccess) {
    (void)hipFree(m->prep); delete m; return CNF_ERR_HIP;
  }
  *out = m;
  return CNF_OK;
}

static void prof_clear(CnfModel* m) {
  for (auto& r : m->prof) {
    if (r.e0) (void)hipEventDestroy(r.e0);
    if (r.e1) (void)hipEventDestroy(r.e1);
    if (r.e2) (void)hipEventDestroy(r.e2);
  }
  m->prof.clear();
}

extern "C" void cnf_model_destroy(CnfModel* m) {
  if (!m) return;
  if (m->prep) (void)hipFree(m->prep);
  if (m->grad_slabs) (void)hipFree(m->grad_slabs);
  if (m->pwl_stats) (void)hipFree(m->pwl_stats);
  for (auto& kv : m->pwl_ws) {
    if (kv.second.tables) (void)hipFree(kv.second.tables);
    for (float* p : kv.second.retired) (void)hipFree(p);
  }
  if (m->prep_event) (void)hipEventDestroy(m->prep_event);
  prof_clear(m);
  delete m;
}

/* Which kernels the most recent compute call of this model ran: a CnfPath value (cnf_common.h).
 * Tests use it to assert that a forced path was really taken; bench.py labels its roofline with it. */
extern "C" int cnf_model_last_path(const CnfModel* m) { return m ? m->last_path : CNF_ERR_INVALID; }

/* Internal (bench.py): with profiling on, the flow entry points record HIP events around their kernels
 * (table path: before the table build, between build and flow kernel, after the flow kernel), at most
 * 4096 launches.  cnf_model_read_profile waits for them and returns the SUMS in milliseconds of the
 * dominant (flow) kernel and of the table build, the number of kernel launches and the samples they
 * processed, then clears the records. */
extern "C" int cnf_model_set_profiling(CnfModel* m, int on) {
  if (!m) return CNF_ERR_INVALID;
  m->profiling = on ? 1 : 0;
  if (!on) prof_clear(m);
  return CNF_OK;
}

extern "C" int cnf_model_read_profile(CnfModel* m, double* flow_ms, double* build_ms, int64_t* launches,
                                      int64_t* samples) {
  if (!m) return CNF_ERR_INVALID;
  double f = 0.0, b = 0.0;
  int64_t n = 0, smp = 0;
  for (auto& r : m->prof) {
    if (hipEventSynchronize(r.e2) != hipSuccess) return CNF_ERR_HIP;
    float ms = 0.f;
    if (r.e0) { if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) return CNF_ERR_HIP; b += ms; }
    if (hipEventElapsedTime(&ms, r.e1, r.e2) != hipSuccess) return CNF_ERR_HIP;
    f += ms; ++n; smp += r.samples;
  }
  prof_clear(m);
  if (flow_ms) *flow_ms = f;
  if (build_ms) *build_ms = b;
  if (launches) *launches = n;
  if (samples) *samples = smp;
  return CNF_OK;
}

// profiling helpers: a record is opened before the (optional) build kernel and closed after the flow kernel
struct ProfScope {
  CnfModel* m; hipStream_t s; CnfModel::ProfRec r; bool on;
  ProfScope(CnfModel* m_, hipStream_t s_, bool with_build, int64_t samples, int path) : m(m_), s(s_), on(false) {
    r.e0 = r.e1 = r.e2 = nullptr; r.samples = samples; r.path = path;
    if (!m->profiling || m->prof.size() >= 4096) return;
    if (hipEventCreate(&r.e1) != hipSuccess || hipEventCreate(&r.e2) != hipSuccess) return;
    if (with_build) { if (hipEventCreate(&r.e0) != hipSuccess) return; (void)hipEventRecord(r.e0, s); }
    else (void)hipEventRecord(r.e1, s);
    on = true;
  }
  void built() { if (on && r.e0) (void)hipEventRecord(r.e1, s); }
  void done() { if (on) { (void)hipEventRecord(r.e2, s); m->prof.push_back(r); on = false; } }
};

/* Internal knob used by the tests and the bench: 1 = hardware transcendentals
 * (default), 0 = ocml expf/logf/sqrtf + IEEE division. */
extern "C" int cnf_model_set_fast_math(CnfModel* m, int on) {
  if (!m) return CNF_ERR_INVALID;
  m->fast_math = on ? 1 : 0;
  return CNF_OK;
}

/* Internal knob: 1 = MFMA conditioner wherever available, 0 = packed-VALU conditioner, 2 = MFMA for launches
 * that leave the chip under-filled (default). */
extern "C" int cnf_model_set_mfma(CnfModel* m, int mode) {
  if (!m || mode < 0 || mode > 2) return CNF_ERR_INVALID;
  m->use_mfma = mode;
  return CNF_OK;
}

/* Internal knob: 1 = piecewise-linear conditioner tables at dim 2 for large launches (default),
 * 2 = for every launch they apply to (tests), 0 = always evaluate the MLP. */
extern "C" int cnf_model_set_pwl(CnfModel* m, int mode) {
  if (!m || mode < 0 || mode > 2) return CNF_ERR_INVALID;
  m->use_pwl = mode;
  return CNF_OK;
}

/* 1 (default): cnf_log_prob / cnf_inverse_logdet (data -> base) carry the knot positions, the offset in the
 * bin and the base term in float64 (cnf_device.h "precise position path"); 0: plain fp32 throughout. */
extern "C" int cnf_model_set_precise(CnfModel* m, int on) {
  if (!m) return CNF_ERR_INVALID;
  m->precise = on ? 1 : 0;
  return CNF_OK;
}

/* Internal knob: wave-per-dimension kernel for base -> data at dim >= 3: 1 = by batch size (default),
 * 2 = always, 0 = never. */
extern "C" int cnf_model_set_dpar(CnfModel* m, int mode) {
  if (!m || mode < 0 || mode > 2) return CNF_ERR_INVALID;
  m->use_dpar = mode;
  return CNF_OK;
}

/* Internal knob: 0 = choose by batch size, 1 / 2 = force samples per lane. */
extern "C" int cnf_model_set_samples_per_lane(CnfModel* m, int spl) {
  if (!m || spl < 0 || spl > 2) return CNF_ERR_INVALID;
  m->force_spl = spl;
  return CNF_OK;
}

extern "C" int cnf_model_set_params(CnfModel* m, const float* params, void* stream) {
  if (!m || !params) return CNF_ERR_INVALID;
  const int K = m->cfg.num_bins;
  const int64_t n_w = m->n_params - (3 * K + 1);
  int blocks = (int)((n_w + 255) / 256);
  if (blocks < 1) blocks = 1;
  if (blocks > 256) blocks = 256;
  blocks += 1;                                  // (+ the block that normalises the `first` spline)
  hipLaunchKernelGGL(prepare_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, m->prep, K,
                     m->n_params, (double)m->cfg.range_min, (double)m->cfg.range_max,
                     (double)m->cfg.min_bin_size, (double)m->cfg.min_knot_slope, m->cfg.dim,
                     m->cfg.num_layers, m->cfg.mlp_num_layers, m->per_layer, m->per_layer_q, m->mfma_off,
                     m->tabd_off, m->cfg.hidden_size, m->cfg.periodized);
  if (m->cfg.periodized && m->cfg.dim > 1)
    hipLaunchKernelGGL(circular_slopes_kernel, dim3((unsigned)(m->cfg.num_layers * (m->cfg.dim - 1))), dim3(64), 0,
                       (hipStream_t)stream, params, m->prep, K, m->cfg.hidden_size, m->cfg.dim, m->cfg.num_layers,
                       m->cfg.mlp_num_layers, m->per_layer);
  if (hipGetLastError() != hipSuccess) return CNF_ERR_HIP;
  // (inside a stream capture the record would become a graph node and leave the event unusable outside the graph:
  //  a captured step is ordered by its own stream)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
  if (cap == hipStreamCaptureStatusNone && hipEventRecord(m->prep_event, (hipStream_t)stream) != hipSuccess) return CNF_ERR_HIP;
  m->prep_stream = stream;
  m->params_set = 1;
  return CNF_OK;
}


constexpr int CNF_MFMA_SMALL_WAVES = 4;        // waves of single-lane work per SIMD up to which use_mfma = 2 picks MFMA

// Two samples per lane (packed fp32) once the batch fills every SIMD with at
// least one wave of sample pairs; one sample per lane below that.
static int samples_per_lane(const CnfModel* m, int64_t B) {
  if (m->force_spl == 1 || m->force_spl == 2) return m->force_spl;
  // the small-launch regime of the MFMA conditioner (launch_flow) is one sample per lane
  if (m->use_mfma == 2 && m->mfma_off > 0 && m->cfg.hidden_size == 16 && m->cfg.num_bins == 5 &&
      B <= (int64_t)m->num_cus * 4 * 64 * CNF_MFMA_SMALL_WAVES) return 1;
  return (m->fast_math && B >= (int64_t)m->num_cus * 4 * 64 * 2) ? 2 : 1;
}

#define CNF_LAUNCH(KERNEL, GRID, LDS, STREAM, ARGS)                          \
  do {                                                                     \
    if (!ensure_lds(KERNEL, LDS)) return CNF_ERR_UNSUPPORTED;              \
    hipLaunchKernelGGL(KERNEL, dim3((unsigned)(GRID)), dim3(TILE), LDS, STREAM, ARGS); \
  } while (0)

// flow_dpar_kernel: D >= 3, base -> data, packed-VALU conditioner, hardware transcendentals.  Chosen (use_dpar
// = 1) while the one-sample-per-lane kernel would leave the chip under-filled.
static int launch_flow_dpar(CnfModel* m, const FlowArgs& a, hipStream_t stream) {
  const int D = m->cfg.dim;
  if (D < 3 || !m->fast_math || m->use_mfma == 1 || !m->use_dpar || m->cfg.periodized) return CNF_ERR_UNSUPPORTED;
  // measured crossover with the one-sample-per-lane kernel (MI355X, D = 3 and D = 10, scripts/exp_dim10.py):
  // between 131 072 and 524 288 samples; 512 samples per CU
  if (m->use_dpar == 1 && a.B > (int64_t)m->num_cus * 512) return CNF_ERR_UNSUPPORTED;
  int nw = D - 1;
  if (nw > 16) nw = 16;
  const int spl = (m->force_spl == 1 || m->force_spl == 2) ? m->force_spl : (a.B >= (int64_t)m->num_cus * 128 ? 2 : 1);
  const int64_t ts = 64 * spl;
  const size_t lds = (size_t)(hdr_floats(m->cfg.num_bins) + (2 * D + nw) * ts) * sizeof(float);
  if (lds > 64 * 1024) return CNF_ERR_UNSUPPORTED;
  int64_t grid = (a.B + ts - 1) / ts;
  const int64_t cap = (int64_t)m->num_cus * 8;
  if (grid > cap) grid = cap;
#define X(HH, KK)                                                                             \
  if (m->cfg.hidden_size == HH && m->cfg.num_bins == KK) {                                    \
    if (!a.gate) m->last_path = CNF_PATH_DPAR;                                                \
    ProfScope ps(m, stream, false, a.B, CNF_PATH_DPAR);                                       \
    if (spl == 2) hipLaunchKernelGGL((flow_dpar_kernel<HH, KK, true, v2f>), dim3((unsigned)grid), dim3(64 * nw), lds, stream, a);   \
    else hipLaunchKernelGGL((flow_dpar_kernel<HH, KK, true, float>), dim3((unsigned)grid), dim3(64 * nw), lds, stream, a);          \
    ps.done();                                                                                \
    return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;                            \
  }
  CNF_KERNEL_CONFIGS(X)
#undef X
  return CNF_ERR_UNSUPPORTED;
}

// LDS of the precise position path beyond the tile: the 2^(-i/32) table and the float64 `first` table
static size_t precise_lds_bytes(int K) { return sizeof(double) * (size_t)(cnf::EXP2_N + hdr_floats(K)); }

template <bool TO_BASE>
static int launch_flow(CnfModel* m, const FlowArgs& a, int spl, hipStream_t stream) {
  const int64_t ts = (int64_t)TILE * spl;
  const int64_t n_tiles = (a.B + ts - 1) / ts;
  int64_t grid = n_tiles;
  const int64_t cap = (int64_t)m->num_cus * 8;
  if (grid > cap) grid = cap;
  constexpr bool PR = TO_BASE;                 // the precise position path exists for the data -> base direction
  if (m->cfg.periodized) {
    // RQSFlow(periodized=True): one sample per lane, hardware transcendentals, plain fp32 positions; sin / cos of
    // the conditioner inputs by ocml.  (fast_math off: the float64 entry points are the exact mode.)
    if (!m->fast_math) return CNF_ERR_UNSUPPORTED;
    const int64_t tiles1 = (a.B + TILE - 1) / TILE;
    const int64_t grid1 = tiles1 < (int64_t)m->num_cus * 8 ? tiles1 : (int64_t)m->num_cus * 8;
    const size_t lds1 = (size_t)(hdr_floats(m->cfg.num_bins) + 2 * a.m.D * TILE) * sizeof(float);
    if (!a.gate) m->last_path = CNF_PATH_MLP1;
#define X(HH, KK)                                                                             \
    if (m->cfg.hidden_size == HH && m->cfg.num_bins == KK) {                                  \
      CNF_LAUNCH((flow_kernel<HH, KK, TO_BASE, true, float, false, false, true>), grid1, lds1, stream, a);  \
      return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;                          \
    }
    CNF_KERNEL_CONFIGS(X)
#undef X
    return CNF_ERR_UNSUPPORTED;
  }
  const bool precise = PR && m->precise;
  const size_t lds = (size_t)(hdr_floats(m->cfg.num_bins) + 2 * a.m.D * ts) * sizeof(float) +
                     (precise ? precise_lds_bytes(m->cfg.num_bins) + sizeof(float) * a.m.D * ts : 0);
  // launches of up to 4 waves of single-lane work per SIMD: the MFMA conditioner, one sample per lane (measured
  // crossover with the packed-VALU kernel, dim 2 and dim 10: profiles/r02_experiments/exp_latency.log)
  const bool small = a.B <= (int64_t)m->num_cus * 4 * 64 * CNF_MFMA_SMALL_WAVES;
  if (m->fast_math && (m->use_mfma == 1 || (m->use_mfma == 2 && small)) && m->mfma_off > 0 &&
      m->cfg.hidden_size == 16 && m->cfg.num_bins == 5) {
    if (!a.gate) m->last_path = CNF_PATH_MFMA;
    ProfScope ps(m, stream, false, a.B, CNF_PATH_MFMA);
    const bool d2 = m->cfg.dim == 2 && spl == 1;        // the reference's per-batch call pattern: its own instantiation
    if (precise) {
      if (spl == 2) CNF_LAUNCH((flow_kernel<16, 5, TO_BASE, true, v2f, true, PR>), grid, lds, stream, a);
      else if (d2) CNF_LAUNCH((flow_kernel<16, 5, TO_BASE, true, float, true, PR, false, 2>), grid, lds, stream, a);
      else CNF_LAUNCH((flow_kernel<16, 5, TO_BASE, true, float, true, PR>), grid, lds, stream, a);
    } else {
      if (spl == 2) CNF_LAUNCH((flow_kernel<16, 5, TO_BASE, true, v2f, true>), grid, lds, stream, a);
      else if (d2) CNF_LAUNCH((flow_kernel<16, 5, TO_BASE, true, float, true, false, false, 2>), grid, lds, stream, a);
      else CNF_LAUNCH((flow_kernel<16, 5, TO_BASE, true, float, true>), grid, lds, stream, a);
    }
    ps.done();
    return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
  }
  if (!a.gate) m->last_path = spl == 2 ? CNF_PATH_MLP2 : CNF_PATH_MLP1;
  ProfScope ps(m, stream, false, a.B, spl == 2 ? CNF_PATH_MLP2 : CNF_PATH_MLP1);
#define X(HH, KK)                                                                             \
  if (m->cfg.hidden_size == HH && m->cfg.num_bins == KK) {                                    \
    if (precise) {                                                                            \
      if (!m->fast_math) CNF_LAUNCH((flow_kernel<HH, KK, TO_BASE, false, float, false, PR>), grid, lds, stream, a);  \
      else if (spl == 2) CNF_LAUNCH((flow_kernel<HH, KK, TO_BASE, true, v2f, false, PR>), grid, lds, stream, a);     \
      else CNF_LAUNCH((flow_kernel<HH, KK, TO_BASE, true, float, false, PR>), grid, lds, stream, a);                 \
    } else {                                                                                  \
      if (!m->fast_math) CNF_LAUNCH((flow_kernel<HH, KK, TO_BASE, false, float>), grid, lds, stream, a);      \
      else if (spl == 2) CNF_LAUNCH((flow_kernel<HH, KK, TO_BASE, true, v2f>), grid, lds, stream, a);         \
      else CNF_LAUNCH((flow_kernel<HH, KK, TO_BASE, true, float>), grid, lds, stream, a);                     \
    }                                                                                         \
    ps.done();                                                                                \
    return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;                            \
  }
  CNF_KERNEL_CONFIGS(X)
#undef X
  return CNF_ERR_UNSUPPORTED;
}

// This stream's table workspace (cnf_model_reserve).  Lookup only: the compute entry points never allocate,
// free or synchronise.  *sets = 0 when the stream has no reservation.
static void pwl_workspace(CnfModel* m, hipStream_t stream, float** tables, int64_t* sets, uint32_t** flag = nullptr,
                          uint32_t* epoch = nullptr) {
  std::lock_guard<std::mutex> lock(m->pwl_mu);
  auto it = m->pwl_ws.find((void*)stream);
  if (it == m->pwl_ws.end()) { *tables = nullptr; *sets = 0; return; }
  *tables = it->second.tables; *sets = it->second.sets;
  // the uniform-condition stamp lives behind the tables; a call that uses it takes a fresh epoch
  if (flag) { *flag = reinterpret_cast<uint32_t*>(it->second.tables + it->second.sets * m->cfg.num_layers * (int64_t)cnf::PWL_TBL);
              *epoch = ++it->second.epoch; }
}

extern "C" int64_t cnf_model_table_bytes(const CnfModel* m) {
  return m ? (int64_t)sizeof(float) * m->cfg.num_layers * cnf::PWL_TBL : 0;
}

extern "C" int cnf_model_reserve(CnfModel* m, void* stream, int64_t n_sets) {
  if (!m || n_sets < 0) return CNF_ERR_INVALID;
  std::lock_guard<std::mutex> lock(m->pwl_mu);
  CnfModel::PwlWorkspace& ws = m->pwl_ws[stream];          // value-initialised on first use
  if (ws.sets >= n_sets && n_sets > 0) return CNF_OK;
  if (ws.tables && n_sets > 0) {
    // Growing: the old block is RETIRED, not freed -- a HIP graph captured on this stream has its address baked into
    // kernel arguments and may be replayed at any later time (a stream synchronisation protects running kernels, not
    // future replays).  Retired blocks live until cnf_model_destroy or an explicit release (n_sets = 0); reservations
    // grow geometrically (FlowEngine.reserve), so they add up to less than the current block.
    ws.retired.push_back(ws.tables);
    ws.tables = nullptr; ws.sets = 0;
  }
  if (n_sets == 0) {      // explicit release: the caller vouches that nothing (no graph either) uses this stream's tables
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return CNF_ERR_HIP;
    if (ws.tables) (void)hipFree(ws.tables);
    for (float* p : ws.retired) (void)hipFree(p);
    m->pwl_ws.erase(stream);
    return CNF_OK;
  }
  if (hipMalloc((void**)&ws.tables, (size_t)cnf_model_table_bytes(m) * (size_t)n_sets + 64) != hipSuccess) {
    ws.tables = nullptr; ws.sets = 0;        // (calls on this stream fall back to the MLP kernels; retired blocks stay)
    return CNF_ERR_NOMEM;
  }
  // the stamp of cond_uniform_kernel starts at 0; epochs count from 1
  if (hipMemset(reinterpret_cast<char*>(ws.tables) + (size_t)cnf_model_table_bytes(m) * (size_t)n_sets, 0, 64) != hipSuccess) {
    (void)hipFree(ws.tables); ws.tables = nullptr; ws.sets = 0;
    return CNF_ERR_HIP;
  }
  ws.sets = n_sets; ws.epoch = 0;
  return CNF_OK;
}

extern "C" int64_t cnf_model_reserved(CnfModel* m, void* stream) {
  if (!m) return CNF_ERR_INVALID;
  float* t; int64_t sets;
  pwl_workspace(m, (hipStream_t)stream, &t, &sets);
  return sets;
}

static const int64_t PWL_MAX_SLICES = 2048;      // slices per build + flow kernel pair

static bool pwl_config_ok(const CnfModel* m) {
  const CnfConfig& g = m->cfg;
  return m->use_pwl && m->fast_math && g.dim == 2 && g.hidden_size == cnf::PWL_H && g.num_bins == 5 &&
         g.mlp_num_layers == 2 && !g.periodized;      // (sin / cos features are not piecewise linear in u)
}

// The piecewise-linear path (cnf_pwl.h): dim 2, H = 16, K = 5, two MLP layers, a condition that is
// uniform over slices of even length, 16-byte aligned points.  Returns CNF_ERR_UNSUPPORTED when the
// launch does not qualify (the caller then runs the MLP kernel).
// `detect`: the condition is per-sample in form (c_block == 1); the caller passes c_block = B here.  The
// uniformity check is enqueued first and the table kernels run only if it finds c uniform; *gate / *gate_epoch
// return the stamp for the MLP kernel the caller enqueues behind them (which runs only if c is NOT uniform).
// Where a base -> data call takes its points from when `in` is null: the cnf_fill_normal stream of `seed`
struct NoiseSrc { uint64_t seed; int64_t first_sample, slice_stride; };

static int run_flow_pwl(CnfModel* m, bool to_base, const float* in, const float* c, int64_t c_block,
                        float* out, float* aux, int aux_mode, int64_t B, hipStream_t stream,
                        bool detect = false, const uint32_t** gate = nullptr, uint32_t* gate_epoch = nullptr,
                        const NoiseSrc* noise = nullptr, const float* built = nullptr, bool in_shared = false) {
  // built: the slices' tables are in the stream's workspace already (cnf_internal_build_tables: one chunk);
  // in_shared: `in` holds ONE slice of points that every slice reads
  if (!pwl_config_ok(m)) return CNF_ERR_UNSUPPORTED;
  const int L = m->cfg.num_layers;
  const bool precise = to_base && m->precise;
  // every row of the L tables in LDS if that fits (L <= 3), else the first PWL_LROWS rows
  auto lds_for = [&](int lrows) {
    return (size_t)(((cnf::hdr_floats(5) + 3) & ~3) + L * cnf::pwl_ltbl(lrows)) * sizeof(float) +
           (precise ? precise_lds_bytes(5) : 0);
  };
  const bool full = lds_for(cnf::PWL_NPIECE) <= 160 * 1024;
  size_t lds = lds_for(full ? cnf::PWL_NPIECE : cnf::PWL_LROWS);
  if (lds > 160 * 1024) return CNF_ERR_UNSUPPORTED;
  const int64_t slice_len = c_block < B ? c_block : B;
  const int64_t n_slices = (B + slice_len - 1) / slice_len;
  if (n_slices > 1 && (slice_len & 1)) return CNF_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) ||
      (reinterpret_cast<uintptr_t>(aux) & 7))
    return CNF_ERR_UNSUPPORTED;
  // Measured (MI355X, 256 x 65 536): the kernel is VALU-bound and runs best at 4 waves per SIMD -- one
  // 1024-thread workgroup per CU 58.9 G samples/s; 6 waves (3 x 512) 56.9; 2 waves 47.8.  Asking for at
  // least 82 KB of LDS keeps a second workgroup off the CU.
  // (profiles/r02_experiments: other workgroup sizes and LDS requests -- 2 / 4 / 6 waves per SIMD -- were all slower)
  const int pwl_threads = cnf::PWL_MAX_THREADS;
  const size_t pwl_min_lds = 82 * 1024;
  const int64_t PWL_TS = 2 * pwl_threads;
  const int64_t tps = (slice_len + PWL_TS - 1) / PWL_TS;
  const int64_t total = n_slices * tps;
  if (total > (1 << 30)) return CNF_ERR_UNSUPPORTED;
  // the tables cost one small kernel per launch: worth it once every CU has a tile,
  // and only while a slice is long enough to amortise building its tables
  // (crossover with the MLP kernel: B / 26 G/s = 20 us of table building + B / 62 G/s  ->  B ~ 0.9 M samples)
  if (m->use_pwl == 1 && (total < 2 * (int64_t)m->num_cus || slice_len < 4 * PWL_TS)) return CNF_ERR_UNSUPPORTED;
  if (pwl_min_lds > lds) lds = pwl_min_lds;
  typedef void (*PwlKernel)(const cnf::PwlArgs);
  constexpr int ALL = cnf::PWL_NPIECE, WIN = cnf::PWL_LROWS;
  // L = 2 (every configuration of the reference) has its own instantiation with the layer loop unrolled
  const bool l2 = full && L == 2;
  const PwlKernel kern =
      precise ? (l2 ? (PwlKernel)cnf::flow_pwl_kernel<5, true, true, true, ALL, 2>
                    : full ? (PwlKernel)cnf::flow_pwl_kernel<5, true, true, true, ALL> : (PwlKernel)cnf::flow_pwl_kernel<5, true, true, true, WIN>)
      : to_base ? (l2 ? (PwlKernel)cnf::flow_pwl_kernel<5, true, true, false, ALL, 2>
                      : full ? (PwlKernel)cnf::flow_pwl_kernel<5, true, true, false, ALL> : (PwlKernel)cnf::flow_pwl_kernel<5, true, true, false, WIN>)
      : noise ? (l2 ? (PwlKernel)cnf::flow_pwl_kernel<5, false, true, false, ALL, 2, true>          // (base noise drawn in the kernel)
                    : full ? (PwlKernel)cnf::flow_pwl_kernel<5, false, true, false, ALL, 0, true> : (PwlKernel)cnf::flow_pwl_kernel<5, false, true, false, WIN, 0, true>)
                : (l2 ? (PwlKernel)cnf::flow_pwl_kernel<5, false, true, false, ALL, 2>
                      : full ? (PwlKernel)cnf::flow_pwl_kernel<5, false, true, false, ALL> : (PwlKernel)cnf::flow_pwl_kernel<5, false, true, false, WIN>);
  if (!ensure_lds(kern, lds)) return CNF_ERR_UNSUPPORTED;
  // at most PWL_MAX_SLICES slices per kernel pair: the workspace stays bounded (2 048 x L x 48 KB) however many
  // slices a call has
  int64_t chunk = n_slices < PWL_MAX_SLICES ? n_slices : PWL_MAX_SLICES;
  float* tables = nullptr;
  {
    // the stream's reservation decides: none (or one too small to keep every CU busy) -> the MLP kernel
    int64_t sets = 0;
    pwl_workspace(m, stream, &tables, &sets);
    if (sets < chunk) chunk = sets;
    if (chunk < 1 || (built && (chunk < n_slices || built != tables))) return CNF_ERR_UNSUPPORTED;
    if (m->use_pwl == 1 && chunk < n_slices && chunk * tps < (int64_t)m->num_cus) return CNF_ERR_UNSUPPORTED;
  }
  uint32_t* flag = nullptr;
  uint32_t epoch = 0;
  if (detect) {
    int64_t sets = 0;
    pwl_workspace(m, stream, &tables, &sets, &flag, &epoch);
    int64_t g = (B + 4095) / 4096;
    if (g > 4 * (int64_t)m->num_cus) g = 4 * (int64_t)m->num_cus;
    hipLaunchKernelGGL(cnf::cond_uniform_kernel, dim3((unsigned)g), dim3(256), 0, stream, c, B, flag, epoch);
    *gate = flag; *gate_epoch = epoch;
  }
  const double sp_offset = log(exp(1.0 - (double)m->cfg.min_knot_slope) - 1.0);
  m->last_path = detect ? CNF_PATH_DETECT : CNF_PATH_TABLES;
  for (int64_t s0 = 0; s0 < n_slices; s0 += chunk) {
    const int64_t ns = n_slices - s0 < chunk ? n_slices - s0 : chunk;
    const int64_t first = s0 * slice_len;
    ProfScope ps(m, stream, true, (B - first) < ns * slice_len ? (B - first) : ns * slice_len, CNF_PATH_TABLES);
    if (!built)
      hipLaunchKernelGGL(cnf::pwl_build_kernel, dim3((unsigned)(ns * L)), dim3(512), 0, stream,
                         (const float*)(m->prep + cnf::hdr_floats(5)), m->per_layer, c + s0, 0.0f, L, sp_offset, tables);
    ps.built();
    cnf::PwlArgs a;
    a.m = model_args(m);
    a.in_shared = in_shared ? 1 : 0;
    a.in = in ? in + (in_shared ? 0 : first * 2) : nullptr; a.out = out ? out + first * 2 : nullptr; a.aux = aux ? aux + first : nullptr;
    a.seed = noise ? noise->seed : 0; a.slice_stride = noise ? noise->slice_stride : 0;
    a.first_sample = noise ? noise->first_sample + s0 * noise->slice_stride : 0;
    a.tables = tables;
    a.B = (B - first) < ns * slice_len ? (B - first) : ns * slice_len;
    a.slice_len = slice_len;
    a.n_slices = (int32_t)ns; a.tiles_per_slice = (int32_t)tps; a.aux_mode = aux_mode;
    a.gate = flag; a.gate_epoch = epoch; a.gate_want = 0;        // tables: only if no block stamped a difference
    const int64_t tiles = ns * tps;
    const int64_t want = (int64_t)m->num_cus;
    const int64_t grid = tiles < want ? tiles : want;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(pwl_threads), lds, stream, a);
    ps.done();
  }
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

// base -> data over n_slices slices of slice_len points each, all reading the ONE slice of points `in`, on tables
// already built in the stream's workspace (cnf_kinetic_potential_vjp)
int cnf_internal_flow_shared(CnfModel* m, hipStream_t stream, const float* in, const float* c, int64_t slice_len,
                             int64_t n_slices, const float* tables, float* out) {
  return run_flow_pwl(m, false, in, c, slice_len, out, nullptr, 0, n_slices * slice_len, stream, false, nullptr, nullptr,
                      nullptr, tables, true);
}

int cnf_internal_build_tables(CnfModel* m, hipStream_t stream, const float* c, int64_t n, float** tables) {
  if (!pwl_config_ok(m) || n < 1) return CNF_ERR_UNSUPPORTED;
  int64_t sets = 0;
  pwl_workspace(m, stream, tables, &sets);
  if (sets < n) return CNF_ERR_UNSUPPORTED;
  const int L = m->cfg.num_layers;
  const double sp_offset = log(exp(1.0 - (double)m->cfg.min_knot_slope) - 1.0);
  hipLaunchKernelGGL(cnf::pwl_build_kernel, dim3((unsigned)(n * L)), dim3(512), 0, stream,
                     (const float*)(m->prep + cnf::hdr_floats(5)), m->per_layer, c, 0.0f, L, sp_offset, *tables);
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

static int run_flow(CnfModel* m, bool to_base, const float* in, const float* c, int64_t c_block,
                    float* out, float* aux, int aux_mode, int64_t B, void* stream, const NoiseSrc* noise = nullptr) {
  if (!m || (!in && !noise) || (in && noise) || !c || B < 0 || c_block < 1) return CNF_ERR_INVALID;
  if (noise && (to_base || noise->first_sample < 0 || noise->slice_stride < 0)) return CNF_ERR_INVALID;
  if (noise && c_block == 1 && B > 1 && noise->slice_stride != 1) return CNF_ERR_INVALID;      // per-sample conditions: one run of the stream
  if (!out && !aux) return CNF_ERR_INVALID;
  if (!m->params_set) return CNF_ERR_INVALID;
  if (B == 0) return CNF_OK;
  if (wait_for_params(m, (hipStream_t)stream) != CNF_OK) return CNF_ERR_HIP;
  const uint32_t* gate = nullptr;
  uint32_t gate_epoch = 0;
  if (c_block == 1 && B > 1) {
    // per-sample conditions: uniform in every reference call (one time broadcast to cond[B,1]).  When the launch
    // is one the table path would take as a single slice, enqueue the check + the table kernels + (below) the MLP
    // kernel, gated on the device by the check's result.
    const int r = run_flow_pwl(m, to_base, in, c, B, out, aux, aux_mode, B, (hipStream_t)stream, true, &gate, &gate_epoch, noise);
    if (r != CNF_OK && r != CNF_ERR_UNSUPPORTED) return r;
  } else {
    const int r = run_flow_pwl(m, to_base, in, c, c_block, out, aux, aux_mode, B, (hipStream_t)stream, false, nullptr, nullptr, noise);
    if (r != CNF_ERR_UNSUPPORTED) return r;
  }
  FlowArgs a;
  a.m = model_args(m);
  a.in = in; a.c = c; a.out = out; a.aux = aux;
  a.B = B; a.c_block = c_block;
  a.gate = gate; a.gate_epoch = gate_epoch; a.gate_want = 1;     // MLP kernel: only if a difference was stamped
  a.fd2 = 0; a.fd_h = 0.f; a.fd_inv_dx = 0.f;
  a.seed = noise ? noise->seed : 0; a.first_sample = noise ? noise->first_sample : 0; a.slice_stride = noise ? noise->slice_stride : 0;
  a.aux_mode = aux_mode;
  a.div_magic = m->div_magic;
  int spl = m->fast_math ? samples_per_lane(m, B) : 1;
  // two samples per lane double the LDS tile: fall back when it would not fit
  if (spl == 2 && (size_t)(hdr_floats(m->cfg.num_bins) + 3 * a.m.D * TILE * 2) * sizeof(float) +
                      precise_lds_bytes(m->cfg.num_bins) > 160 * 1024) spl = 1;
  if (c_block >= B) a.c_mode = C_SINGLE;
  else if (c_block == 1) a.c_mode = C_PER_SAMPLE;
  else if (c_block % (TILE * spl) == 0) a.c_mode = C_TILE_UNIFORM;
  else a.c_mode = C_GENERIC;
  if (!to_base) {
    const int r = launch_flow_dpar(m, a, (hipStream_t)stream);
    if (r != CNF_ERR_UNSUPPORTED) return r;
  }
  return to_base ? launch_flow<true>(m, a, spl, (hipStream_t)stream)
                 : launch_flow<false>(m, a, spl, (hipStream_t)stream);
}

extern "C" int cnf_forward_logdet(CnfModel* m, const float* x, const float* c, int64_t c_block,
                                  float* y, float* logdet, int64_t B, void* stream) {
  return run_flow(m, false, x, c, c_block, y, logdet, AUX_LOGDET, B, stream);
}

extern "C" int cnf_inverse_logdet(CnfModel* m, const float* y, const float* c, int64_t c_block,
                                  float* x, float* logdet, int64_t B, void* stream) {
  return run_flow(m, true, y, c, c_block, x, logdet, AUX_LOGDET, B, stream);
}

extern "C" int cnf_log_prob(CnfModel* m, const float* value, const float* c, int64_t c_block,
                            float* logp, int64_t B, void* stream) {
  if (!logp) return CNF_ERR_INVALID;
  return run_flow(m, true, value, c, c_block, nullptr, logp, AUX_LOGPROB, B, stream);
}

extern "C" int cnf_sample_logprob(CnfModel* m, const float* noise, const float* c, int64_t c_block,
                                  float* y, float* logp, int64_t B, void* stream) {
  if (!y) return CNF_ERR_INVALID;
  return run_flow(m, false, noise, c, c_block, y, logp, AUX_LOGPROB, B, stream);
}

extern "C" int cnf_sample_logprob_seeded(CnfModel* m, uint64_t seed, int64_t first_sample, int64_t slice_stride,
                                         const float* c, int64_t c_block, float* y, float* logp, int64_t B,
                                         void* stream) {
  if (!y) return CNF_ERR_INVALID;
  const NoiseSrc noise{seed, first_sample, slice_stride};
  return run_flow(m, false, nullptr, c, c_block, y, logp, AUX_LOGPROB, B, stream, &noise);
}

extern "C" int cnf_fill_normal_threefry(uint32_t key0, uint32_t key1, uint64_t size, uint64_t first_element, int64_t n,
                                        float* out_f32, double* out_f64, void* stream) {
  if (n < 0 || (n > 0 && !out_f32 && !out_f64) || first_element + (uint64_t)n > size || size > 0x7fffffffull)
    return CNF_ERR_INVALID;        // (2 size 32-bit counters: jax's single-block case)
  if (n == 0) return CNF_OK;
  int64_t grid = (n + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(fill_normal_threefry_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, key0, key1,
                     size, first_element, n, out_f32, out_f64);
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

extern "C" int cnf_logprob_fd(CnfModel* m, const float* pts, const float* c, int64_t c_block, float dx,
                              float* score, int64_t B, void* stream) {
  if (!m || !pts || !c || !score || B < 0 || c_block < 1 || !(dx > 0.f)) return CNF_ERR_INVALID;
  if (!m->params_set) return CNF_ERR_INVALID;
  if (B == 0) return CNF_OK;
  const int D = m->cfg.dim;
  if (B * 2 * D >= ((int64_t)1 << 40)) return CNF_ERR_INVALID;
  if (wait_for_params(m, (hipStream_t)stream) != CNF_OK) return CNF_ERR_HIP;
  FlowArgs a;
  a.m = model_args(m);
  a.in = pts; a.c = c; a.out = nullptr; a.aux = score;
  a.B = B * 2 * D;                              // evaluation points
  a.c_block = c_block;
  a.aux_mode = AUX_LOGPROB;
  a.div_magic = m->div_magic;
  a.gate = nullptr; a.gate_epoch = 0; a.gate_want = 0;
  a.fd2 = 2 * D; a.fd_h = 0.5f * dx; a.fd_inv_dx = 1.0f / dx;
  a.seed = 0; a.first_sample = 0; a.slice_stride = 0;
  a.c_mode = c_block >= B ? C_SINGLE : C_GENERIC;
  int spl = m->fast_math ? samples_per_lane(m, a.B) : 1;
  if (spl == 2 && (size_t)(hdr_floats(m->cfg.num_bins) + 2 * D * TILE * 2) * sizeof(float) > 160 * 1024) spl = 1;
  // plain fp32: the difference of two nearby log_prob values cancels what the precise position path would fix
  const int precise = m->precise;
  m->precise = 0;
  const int r = launch_flow<true>(m, a, spl, (hipStream_t)stream);
  m->precise = precise;
  return r;
}

extern "C" int cnf_fill_normal(uint64_t seed, uint64_t first_element, int64_t n, float* out,
                               void* stream) {
  if (n < 0 || (n > 0 && !out)) return CNF_ERR_INVALID;
  if (n == 0) return CNF_OK;
  const uint64_t n_blk = ((first_element + (uint64_t)n - 1) >> 2) - (first_element >> 2) + 1;
  uint64_t grid = (n_blk + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(fill_normal_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, seed,
                     first_element, n, out, (const uint64_t*)nullptr);
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

// ---- the random inputs of a training step drawn from a key in DEVICE memory ---------------------------------------
// state: uint64[2] on the device = { step count, key }.  The caller writes the key (one 8-byte copy) before a step;
// every draw below reads it on the device, so the whole step -- draws, loss, gradient, Adam -- can be captured into
// a HIP graph once and replayed with a new key each time.  Streams of one key: the normal stream of
// cnf_fill_normal (Philox counter words 2, 3 = 0, 0), uniforms (word 2 = 1) and 3-bit integers (word 2 = 2).
namespace cnf {
__global__ void step_begin_kernel(uint64_t* state) { if (threadIdx.x == 0 && blockIdx.x == 0) state[0] += 1; }

// out[i] = scale * u, u = 24-bit uniform in [0, 1) from word (first + i) & 3 of block (first + i) >> 2 of stream 1
__global__ void fill_uniform_kernel(const uint64_t* __restrict__ state, uint64_t first, int64_t n, float scale,
                                    float* __restrict__ out) {
  const uint64_t key = state[1];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t e = first + (uint64_t)i;
    uint32_t u[4];
    philox4x32((uint32_t)(e >> 2), (uint32_t)(e >> 34), 1u, 0u, (uint32_t)key, (uint32_t)(key >> 32), u);
    out[i] = scale * ((float)(u[e & 3] >> 8) * (1.0f / 16777216.0f));
  }
}

// The 8-mode mixture source of kl_loss_fn (applications.py:34-71) for n samples of dim 2: out[i] = z[i] + centre of
// component (first_sample + i), the component = the top 3 bits of word e & 3 of block e >> 2 of stream 2
__global__ void mixture_source_kernel(const uint64_t* __restrict__ state, uint64_t first_sample, int64_t n,
                                      const float* __restrict__ z, float* __restrict__ out, int32_t* __restrict__ comp_out) {
  constexpr float R = 5.0f;
  const float cx[8] = {0.0f, 1.0f, 0.0f, -1.0f, 0.6f, 0.6f, -0.6f, -0.6f};
  const float cy[8] = {1.0f, 0.0f, -1.0f, 0.0f, 0.8f, -0.8f, -0.8f, 0.8f};
  const uint64_t key = state[1];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t e = first_sample + (uint64_t)i;
    uint32_t u[4];
    philox4x32((uint32_t)(e >> 2), (uint32_t)(e >> 34), 2u, 0u, (uint32_t)key, (uint32_t)(key >> 32), u);
    const int k = (int)(u[e & 3] >> 29);
    float mx = 0.0f, my = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { mx = k == j ? cx[j] : mx; my = k == j ? cy[j] : my; }
    if (out) { out[2 * i] = z[2 * i] + R * mx; out[2 * i + 1] = z[2 * i + 1] + R * my; }
    if (comp_out) comp_out[i] = k;
  }
}
}  // namespace cnf

extern "C" int cnf_step_begin(uint64_t* state, void* stream) {
  if (!state) return CNF_ERR_INVALID;
  hipLaunchKernelGGL(cnf::step_begin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state);
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

extern "C" int cnf_fill_normal_dev(const uint64_t* state, uint64_t first_element, int64_t n, float* out, void* stream) {
  if (!state || n < 0 || (n > 0 && !out)) return CNF_ERR_INVALID;
  if (n == 0) return CNF_OK;
  const uint64_t n_blk = ((first_element + (uint64_t)n - 1) >> 2) - (first_element >> 2) + 1;
  uint64_t grid = (n_blk + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(fill_normal_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (uint64_t)0,
                     first_element, n, out, state);
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

extern "C" int cnf_fill_uniform_dev(const uint64_t* state, uint64_t first, int64_t n, float scale, float* out, void* stream) {
  if (!state || n < 0 || (n > 0 && !out)) return CNF_ERR_INVALID;
  if (n == 0) return CNF_OK;
  int64_t grid = (n + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(cnf::fill_uniform_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, state, first, n, scale, out);
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

extern "C" int cnf_mixture_source_dev(const uint64_t* state, uint64_t first_sample, int64_t n, const float* z, float* out,
                                      int32_t* comp, void* stream) {
  if (!state || n < 0 || (n > 0 && !out && !comp) || (out && !z)) return CNF_ERR_INVALID;
  if (n == 0) return CNF_OK;
  int64_t grid = (n + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(cnf::mixture_source_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, state, first_sample,
                     n, z, out, comp);
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}


// The loss terms on the conditioner tables (loss_pwl_kernel): same qualification as run_flow_pwl.
static int loss_terms_pwl(CnfModel* m, const CnfLossSpec* spec, const float* pts, int64_t slice_stride,
                          uint64_t seed, int64_t first_sample, const float* t, int64_t n_slices, int64_t B,
                          double* sums, hipStream_t stream) {
  if (!pwl_config_ok(m)) return CNF_ERR_UNSUPPORTED;
  const int L = m->cfg.num_layers;
  const int kind = spec->kind;
  const bool kin = kind <= CNF_TERM_FLOW_MATCHING;
  const int n_sets = kind == CNF_TERM_KINETIC ? 2 : (kin ? 3 : 1);
  const int threads = cnf::PWL_MAX_THREADS;
  const int64_t ts = 2 * threads;
  const size_t lds = (size_t)(((cnf::hdr_floats(5) + 3) & ~3) + n_sets * L * cnf::PWL_LTBL + 2 * ts) * sizeof(float);
  if (lds > 160 * 1024) return CNF_ERR_UNSUPPORTED;
  const int64_t tps = (B + ts - 1) / ts;
  const int64_t total = n_slices * tps;
  if (total > (1 << 30)) return CNF_ERR_UNSUPPORTED;
  if (m->use_pwl == 1 && (total < 2 * (int64_t)m->num_cus || B < 4 * ts)) return CNF_ERR_UNSUPPORTED;
  if (!ensure_lds(cnf::loss_pwl_kernel<5, true>, lds)) return CNF_ERR_UNSUPPORTED;
  int64_t chunk = n_slices < PWL_MAX_SLICES ? n_slices : PWL_MAX_SLICES;
  float* tables = nullptr;
  {
    int64_t sets = 0;
    pwl_workspace(m, stream, &tables, &sets);
    if (sets / n_sets < chunk) chunk = sets / n_sets;
    if (chunk < 1) return CNF_ERR_UNSUPPORTED;
    if (m->use_pwl == 1 && chunk < n_slices && chunk * tps < (int64_t)m->num_cus) return CNF_ERR_UNSUPPORTED;
  }
  const double sp_offset = log(exp(1.0 - (double)m->cfg.min_knot_slope) - 1.0);
  m->last_path = CNF_PATH_LOSS_TABLES;
  for (int64_t s0 = 0; s0 < n_slices; s0 += chunk) {
    const int64_t ns = n_slices - s0 < chunk ? n_slices - s0 : chunk;
    const int64_t set_stride = ns * L * (int64_t)cnf::PWL_TBL;
    for (int s = 0; s < n_sets; ++s) {          // conditions t - dt/2, t + dt/2, t (kinetic kinds) or t
      const float off = !kin ? 0.0f : (s == 0 ? -0.5f * spec->dt : (s == 1 ? 0.5f * spec->dt : 0.0f));
      hipLaunchKernelGGL(cnf::pwl_build_kernel, dim3((unsigned)(ns * L)), dim3(512), 0, stream,
                         (const float*)(m->prep + cnf::hdr_floats(5)), m->per_layer, t + s0, off, L, sp_offset,
                         tables + s * set_stride);
    }
    cnf::LossPwlArgs a;
    a.m = model_args(m); a.spec = *spec; a.t = t + s0; a.sums = sums + s0; a.tables = tables;
    // slice s of the chunk is slice s0 + s of the call: its points / stream positions start s0 * stride later
    a.pts = pts ? pts + s0 * slice_stride * 2 : nullptr;
    a.B = B; a.n_slices = ns; a.pts_slice_stride = slice_stride;
    a.seed = seed; a.first_sample = first_sample + s0 * slice_stride;
    a.n_sets = n_sets; a.tiles_per_slice = (int32_t)tps;
    const int64_t tiles = ns * tps;
    const int64_t grid = tiles < m->num_cus ? tiles : m->num_cus;
    hipLaunchKernelGGL((cnf::loss_pwl_kernel<5, true>), dim3((unsigned)grid), dim3(threads), lds, stream, a);
  }
  return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

static int loss_terms_impl(CnfModel* m, const CnfLossSpec* spec, const float* pts, int64_t slice_stride,
                           uint64_t seed, int64_t first_sample, const float* t, int64_t n_slices, int64_t B,
                           double* sums, void* stream_) {
  if (!m || !spec || !t || !sums || n_slices < 0 || B < 0 || slice_stride < 0 || first_sample < 0) return CNF_ERR_INVALID;
  if (!m->params_set) return CNF_ERR_INVALID;
  if (spec->kind < CNF_TERM_KINETIC || spec->kind > CNF_TERM_NEG_LOGPROB) return CNF_ERR_INVALID;
  if (m->cfg.periodized) return CNF_ERR_UNSUPPORTED;      // flow functions only (include/cnf_ot_amd.h: CnfConfig)
  const int D = m->cfg.dim;
  if (spec->kind <= CNF_TERM_FLOW_MATCHING && !(spec->dt > 0.f)) return CNF_ERR_INVALID;
  if ((spec->kind == CNF_TERM_KINETIC_SCORE || spec->kind == CNF_TERM_FLOW_MATCHING) && !(spec->dx > 0.f))
    return CNF_ERR_INVALID;
  if (spec->kind == CNF_TERM_FLOW_MATCHING) {
    // the reference raises for these (applications.py:359,365); SMILE is 2-D by construction
    if ((spec->subtype == CNF_DRIFT_SMILE || spec->subtype == CNF_DRIFT_NONGRADIENT) && D != 2) return CNF_ERR_INVALID;
    if (spec->subtype == CNF_DRIFT_LORENZ && D != 3) return CNF_ERR_INVALID;
    if (spec->subtype < CNF_DRIFT_OU || spec->subtype > CNF_DRIFT_LORENZ) return CNF_ERR_INVALID;
  }
  if (spec->kind == CNF_TERM_POTENTIAL && (spec->subtype < CNF_POT_QUADRATIC || spec->subtype > CNF_POT_OBSTACLE))
    return CNF_ERR_INVALID;
  if (spec->kind == CNF_TERM_REVERSE_KL && (!(spec->T > 0.f) || !(spec->beta > 0.f))) return CNF_ERR_INVALID;
  hipStream_t stream = (hipStream_t)stream_;
  if (n_slices == 0) return CNF_OK;
  if (hipMemsetAsync(sums, 0, sizeof(double) * (size_t)n_slices, stream) != hipSuccess) return CNF_ERR_HIP;
  if (B == 0) return CNF_OK;
  if (wait_for_params(m, stream) != CNF_OK) return CNF_ERR_HIP;
  {
    const int r = loss_terms_pwl(m, spec, pts, slice_stride, seed, first_sample, t, n_slices, B, sums, stream);
    if (r != CNF_ERR_UNSUPPORTED) return r;
  }
  m->last_path = CNF_PATH_LOSS_MLP;

  LossArgs a;
  a.m = model_args(m); a.spec = *spec; a.pts = pts; a.t = t; a.sums = sums;
  a.B = B; a.n_slices = n_slices; a.pts_slice_stride = slice_stride;
  a.div_magic = m->div_magic; a.seed = seed; a.first_sample = first_sample;
  // five D x TS buffers: keep a workgroup under ~64 KB of LDS
  int spl = (m->fast_math && n_slices * B >= (int64_t)m->num_cus * 4 * 64 * 2) ? 2 : 1;
  if (m->force_spl == 1 || m->force_spl == 2) spl = m->fast_math ? m->force_spl : 1;
  if (spl == 2 && (size_t)(5 * D * TILE * 2) * sizeof(float) > 64 * 1024) spl = 1;
  const int64_t ts = (int64_t)TILE * spl;
  const size_t lds = (size_t)(hdr_floats(m->cfg.num_bins) + 5 * D * ts) * sizeof(float);
  if (lds > 160 * 1024) return CNF_ERR_UNSUPPORTED;
  int64_t grid = ((B + ts - 1) / ts) * n_slices;
  const int64_t cap = (int64_t)m->num_cus * 8;
  if (grid > cap) grid = cap;
#define X(HH, KK)                                                                             \
  if (m->cfg.hidden_size == HH && m->cfg.num_bins == KK) {                                    \
    if (!m->fast_math) CNF_LAUNCH((loss_kernel<HH, KK, false, float>), grid, lds, stream, a);                 \
    else if (spl == 2) CNF_LAUNCH((loss_kernel<HH, KK, true, v2f>), grid, lds, stream, a);                    \
    else CNF_LAUNCH((loss_kernel<HH, KK, true, float>), grid, lds, stream, a);                                \
    return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;                            \
  }
  CNF_KERNEL_CONFIGS(X)
#undef X
  return CNF_ERR_UNSUPPORTED;
}

extern "C" int cnf_loss_terms(CnfModel* m, const CnfLossSpec* spec, const float* pts, int pts_shared,
                              const float* t, int64_t n_slices, int64_t B, double* sums, void* stream) {
  if (!pts) return CNF_ERR_INVALID;
  return loss_terms_impl(m, spec, pts, pts_shared ? 0 : B, 0, 0, t, n_slices, B, sums, stream);
}

extern "C" int cnf_loss_terms_seeded(CnfModel* m, const CnfLossSpec* spec, uint64_t seed, int64_t first_sample,
                                     int64_t slice_stride, const float* t, int64_t n_slices, int64_t B,
                                     double* sums, void* stream) {
  if (spec && spec->kind == CNF_TERM_NEG_LOGPROB) return CNF_ERR_INVALID;   // that term takes data points
  return loss_terms_impl(m, spec, nullptr, slice_stride, seed, first_sample, t, n_slices, B, sums, stream);
}

// ---- float64 instantiation: the reference's own dtype (solvers.py:23) ---------
// Exact-mode entry points: double IO, double table/constants, ocml math, one
// sample per lane.  Parameters stay the float32 vector given to
// cnf_model_set_params (each weight is widened exactly).
template <bool TO_BASE>
static int launch_flow_f64(CnfModel* m, const FlowArgsD& a, hipStream_t stream) {
  const int64_t n_tiles = (a.B + TILE - 1) / TILE;
  int64_t grid = n_tiles;
  const int64_t cap = (int64_t)m->num_cus * 8;
  if (grid > cap) grid = cap;
  const size_t lds = (size_t)(hdr_floats(m->cfg.num_bins) + 2 * a.m.D * TILE) * sizeof(double);
#define X(HH, KK)                                                                    \
  if (m->cfg.hidden_size == HH && m->cfg.num_bins == KK) {                           \
    if (m->cfg.periodized) CNF_LAUNCH((flow_kernel<HH, KK, TO_BASE, false, double, false, false, true>), grid, lds, stream, a); \
    else CNF_LAUNCH((flow_kernel<HH, KK, TO_BASE, false, double>), grid, lds, stream, a); \
    return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;                   \
  }
  CNF_KERNEL_CONFIGS(X)
#undef X
  return CNF_ERR_UNSUPPORTED;
}

static int run_flow_f64(CnfModel* m, bool to_base, const double* in, const double* c, int64_t c_block,
                        double* out, double* aux, int aux_mode, int64_t B, void* stream) {
  if (!m || !in || !c || B < 0 || c_block < 1) return CNF_ERR_INVALID;
  if (!out && !aux) return CNF_ERR_INVALID;
  if (!m->params_set) return CNF_ERR_INVALID;
  if (B == 0) return CNF_OK;
  if (wait_for_params(m, (hipStream_t)stream) != CNF_OK) return CNF_ERR_HIP;
  m->last_path = CNF_PATH_F64;
  FlowArgsD a;
  a.m = model_args(m);
  a.in = in; a.c = c; a.out = out; a.aux = aux;
  a.B = B; a.c_block = c_block; a.aux_mode = aux_mode; a.div_magic = m->div_magic;
  a.gate = nullptr; a.gate_epoch = 0; a.gate_want = 0;
  a.fd2 = 0; a.fd_h = 0.0; a.fd_inv_dx = 0.0;
  a.seed = 0; a.first_sample = 0; a.slice_stride = 0;
  if (c_block >= B) a.c_mode = C_SINGLE;
  else if (c_block == 1) a.c_mode = C_PER_SAMPLE;
  else if (c_block % TILE == 0) a.c_mode = C_TILE_UNIFORM;
  else a.c_mode = C_GENERIC;
  return to_base ? launch_flow_f64<true>(m, a, (hipStream_t)stream) : launch_flow_f64<false>(m, a, (hipStream_t)stream);
}

extern "C" int cnf_forward_logdet_f64(CnfModel* m, const double* x, const double* c, int64_t c_block, double* y,
                                      double* logdet, int64_t B, void* stream) {
  return run_flow_f64(m, false, x, c, c_block, y, logdet, AUX_LOGDET, B, stream);
}
extern "C" int cnf_inverse_logdet_f64(CnfModel* m, const double* y, const double* c, int64_t c_block, double* x,
                                      double* logdet, int64_t B, void* stream) {
  return run_flow_f64(m, true, y, c, c_block, x, logdet, AUX_LOGDET, B, stream);
}
extern "C" int cnf_log_prob_f64(CnfModel* m, const double* value, const double* c, int64_t c_block, double* logp,
                                int64_t B, void* stream) {
  if (!logp) return CNF_ERR_INVALID;
  return run_flow_f64(m, true, value, c, c_block, nullptr, logp, AUX_LOGPROB, B, stream);
}
extern "C" int cnf_sample_logprob_f64(CnfModel* m, const double* noise, const double* c, int64_t c_block, double* y,
                                      double* logp, int64_t B, void* stream) {
  if (!y) return CNF_ERR_INVALID;
  return run_flow_f64(m, false, noise, c, c_block, y, logp, AUX_LOGPROB, B, stream);
}
