// cnf_device.h -- device-side building blocks of the conditional RQS flow for
// gfx950 (CDNA4): prepared `first` table layout, spline evaluation from the
// table (LDS gather) and from per-lane conditioner output (registers), and the
// time-conditioned MLP conditioner with wave-uniform weights.
//
// Reference behaviour restated (never copied; the reference is Python/JAX):
//   spline       distrax.RationalQuadraticSpline, call site flows.py:124-132
//                (algorithm: SURVEY.md Appendix A)
//   conditioner  cnf_ot/models/flows.py:46-86
//   coupling     cnf_ot/models/autoregressive.py:76-136
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cnf {

// Pointer into the AMDGPU constant address space (4).  Global and constant
// addresses coincide; a load through this type from a wave-uniform address is
// always selected as a scalar-cache load (s_load_dword*), so the value lands
// in SGPRs and feeds the per-lane FMAs as a scalar operand.  (Through a plain
// global pointer hipcc falls back to per-lane global_load as soon as the
// kernel also stores to global memory, because it cannot prove the weights
// are not clobbered.)  The prepared buffer is never written by a flow kernel.
typedef const float __attribute__((address_space(4)))* uniform_ptr;

__device__ __forceinline__ uniform_ptr as_uniform(const float* p) {
  return (uniform_ptr)p;
}

// Constant-address-space loads are `invariant`: LICM hoists every weight load
// of the kernel out of the tile loop (1 184 SGPRs at D=2 -> spilled to VGPR
// lanes).  Passing the pointer through an empty asm makes it opaque per use, so
// the loads stay where they are written; it also pins the pointer in SGPRs.
__device__ __forceinline__ uniform_ptr launder(uniform_ptr p) {
  asm volatile("" : "+s"(p));
  return p;
}

// ---------------------------------------------------------------------------
// Prepared model buffer (device, float32), written by prepare_kernel:
//   [0, K*ROW)                 K rows of the `first` spline, one per bin
//   [KNOT_X, +K+1) [KNOT_Y, +K+1)   knot positions for the bin search
//   [TAIL, +6)                 d_lo, d_hi, log d_lo, log d_hi, 1/d_lo, 1/d_hi
//   [hdr_floats(K), ...)       conditioner weights, flat layout of the C ABI
// The `first` spline is shared by every layer and ignores c (flows.py:47-55,
// autoregressive.py:88-92), so it is normalised ONCE per parameter set, in
// float64, instead of once per sample per layer.
// ---------------------------------------------------------------------------
constexpr int ROW = 12;
enum { R_X0 = 0, R_Y0, R_BW, R_BH, R_IBW, R_IBH, R_S, R_ST, R_D0, R_D1, R_L2S, R_PAD };

__host__ __device__ constexpr int knot_x_off(int K) { return K * ROW; }
__host__ __device__ constexpr int knot_y_off(int K) { return K * ROW + (K + 1); }
__host__ __device__ constexpr int tail_off(int K) { return K * ROW + 2 * (K + 1); }
__host__ __device__ constexpr int hdr_floats(int K) { return (tail_off(K) + 6 + 3) & ~3; }
enum { T_DLO = 0, T_DHI, T_LOG_DLO, T_LOG_DHI, T_INV_DLO, T_INV_DHI };

struct SplineConsts {
  float lo, hi;        // range_min, range_max
  float span_eff;      // (hi - lo) - K * min_bin
  float min_bin;
  float min_slope;
  float sp_offset;     // log(exp(1 - min_slope) - 1)
};

// ---------------------------------------------------------------------------
// Math policy.  FAST=false: ocml expf/logf/sqrtf and IEEE division.
// FAST=true: hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32 /
// v_sqrt_f32, 1 ulp each) with the exp argument scaled in two pieces so the
// result keeps ~1 ulp for |x| up to ~80.
// ---------------------------------------------------------------------------
template <bool FAST> struct Math;

template <> struct Math<false> {
  static __device__ __forceinline__ float exp(float x) { return expf(x); }
  static __device__ __forceinline__ float log(float x) { return logf(x); }
  static __device__ __forceinline__ float rcp(float x) { return 1.0f / x; }
  static __device__ __forceinline__ float div(float a, float b) { return a / b; }
  static __device__ __forceinline__ float sqrt(float x) { return sqrtf(x); }
};

template <> struct Math<true> {
  static __device__ __forceinline__ float exp(float x) {
    const float L2E_HI = 1.44269502162933349609375f;     // fl(log2 e)
    const float L2E_LO = 1.925963033500011e-08f;         // log2 e - fl(log2 e)
    float hi = x * L2E_HI;
    float lo = fmaf(x, L2E_HI, -hi) + x * L2E_LO;        // exact product tail
    float r = __builtin_amdgcn_exp2f(hi);
    return fmaf(r, lo * 0.693147182464599609375f, r);    // 2^(hi+lo)
  }
  static __device__ __forceinline__ float log(float x) {
    return __builtin_amdgcn_logf(x) * 0.693147182464599609375f;
  }
  static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
  static __device__ __forceinline__ float div(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    float q = a * r;
    return fmaf(fmaf(-b, q, a), r, q);                   // one Newton step
  }
  static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
};

__device__ __forceinline__ float clip01(float z) { return fminf(fmaxf(z, 0.0f), 1.0f); }

// softplus(t + offset) + m  (distrax _normalize_knot_slopes)
template <bool FAST>
__device__ __forceinline__ float knot_slope(float t, const SplineConsts& sc) {
  float v = t + sc.sp_offset;
  float e = Math<FAST>::exp(-fabsf(v));
  // log1p(e), e in (0,1]: log(1+e) loses nothing that matters in absolute
  // terms (the slope is O(1)); keep the small-e branch exact to first order.
  // (both forms are evaluated and selected: a branch here splits the wave's
  // straight-line code and lets the compiler sink work into it)
  const float l_log = Math<FAST>::log(1.0f + e);
  const float l_ser = fmaf(-0.5f * e, e, e);
  const float l = e < 1e-4f ? l_ser : l_log;
  return fmaxf(v, 0.0f) + l + sc.min_slope;
}

// Shared tail of both directions: from the selected bin to (out, logdet).
// INV=false: distrax _rational_quadratic_spline_fwd; INV=true: ..._inv.
template <bool INV, bool FAST>
__device__ __forceinline__ void rqs_bin_eval(float v, float x0, float y0, float bw, float bh,
                                             float ibw, float ibh, float s, float st,
                                             float d0, float d1, float l2s,
                                             float& out, float& ld) {
  using M = Math<FAST>;
  float z;
  if (INV) {
    float w = clip01((v - y0) * ibh);
    float c = -s * w;
    float b = d0 - st * w;
    float a = s - b;
    float disc = fmaf(b, b, -4.0f * a * c);
    z = clip01(M::div(-2.0f * c, b + M::sqrt(disc)));
    out = fmaf(bw, z, x0);
  } else {
    z = clip01((v - x0) * ibw);
  }
  float sq_z = z * z;
  float z1mz = z - sq_z;
  float omz = 1.0f - z;
  float den = fmaf(st, z1mz, s);
  float iden = M::rcp(den);
  if (!INV) out = fmaf(bh * fmaf(s, sq_z, d0 * z1mz), iden, y0);
  float num2 = fmaf(d1, sq_z, fmaf(2.0f * s, z1mz, d0 * omz * omz));
  // 2 log s + log(num2) - 2 log(den) = l2s + log(num2 / den^2)
  float ldf = l2s + M::log(num2 * iden * iden);
  ld = INV ? -ldf : ldf;
}

// ---------------------------------------------------------------------------
// Spline of the shared `first` parameters: per-lane bin index, LDS row gather.
// `tab` points at the prepared header staged in LDS.
// ---------------------------------------------------------------------------
template <int K, bool INV, bool FAST>
__device__ __forceinline__ void table_spline(const float* tab, float v, const SplineConsts& sc,
                                             float& out, float& ld) {
  const float* pos = tab + (INV ? knot_y_off(K) : knot_x_off(K));
  int k = 0;
#pragma unroll
  for (int j = 1; j < K; ++j) k += (v >= pos[j]) ? 1 : 0;
  const float4* row = reinterpret_cast<const float4*>(tab + k * ROW);
  float4 r0 = row[0], r1 = row[1], r2 = row[2];
  rqs_bin_eval<INV, FAST>(v, r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, out, ld);
  const float* tl = tab + tail_off(K);
  if (v <= sc.lo) {          // linear tails (rare: |v| >= 10)
    out = INV ? fmaf(v - sc.lo, tl[T_INV_DLO], sc.lo) : fmaf(v - sc.lo, tl[T_DLO], sc.lo);
    ld = INV ? -tl[T_LOG_DLO] : tl[T_LOG_DLO];
  }
  if (v >= sc.hi) {
    out = INV ? fmaf(v - sc.hi, tl[T_INV_DHI], sc.hi) : fmaf(v - sc.hi, tl[T_DHI], sc.hi);
    ld = INV ? -tl[T_LOG_DHI] : tl[T_LOG_DHI];
  }
}

// ---------------------------------------------------------------------------
// Spline whose 3K+1 parameters were just produced by the conditioner, one set
// per lane, in registers.  The bin is selected with a compare/select chain
// over the running knot (registers cannot be indexed per lane); only the two
// slopes of the selected bin are normalised (2 softplus instead of K+1).
// ---------------------------------------------------------------------------
template <int K, bool INV, bool FAST>
__device__ __forceinline__ void cond_spline(const float (&th)[3 * K + 1], float v,
                                            const SplineConsts& sc, float& out, float& ld) {
  using M = Math<FAST>;
  float mw = th[0], mh = th[K];
#pragma unroll
  for (int k = 1; k < K; ++k) { mw = fmaxf(mw, th[k]); mh = fmaxf(mh, th[K + k]); }
  float ew[K], eh[K];
  float sw = 0.0f, sh = 0.0f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    ew[k] = M::exp(th[k] - mw);
    eh[k] = M::exp(th[K + k] - mh);
    sw += ew[k];
    sh += eh[k];
  }
  const float aw = sc.span_eff * M::rcp(sw), ah = sc.span_eff * M::rcp(sh);
  float px = sc.lo, py = sc.lo;                 // running knot k
  float wk = fmaf(ew[0], aw, sc.min_bin), hk = fmaf(eh[0], ah, sc.min_bin);
  float x0 = px, y0 = py, bw = wk, bh = hk, t0 = th[2 * K], t1 = th[2 * K + 1];
#pragma unroll
  for (int k = 1; k < K; ++k) {
    px += wk;
    py += hk;
    if (k == K - 1) { wk = sc.hi - px; hk = sc.hi - py; }   // last knot is exactly hi
    else { wk = fmaf(ew[k], aw, sc.min_bin); hk = fmaf(eh[k], ah, sc.min_bin); }
    const bool ge = INV ? (v >= py) : (v >= px);
    x0 = ge ? px : x0; y0 = ge ? py : y0;
    bw = ge ? wk : bw; bh = ge ? hk : bh;
    t0 = ge ? th[2 * K + k] : t0; t1 = ge ? th[2 * K + k + 1] : t1;
  }
  const float d0 = knot_slope<FAST>(t0, sc), d1 = knot_slope<FAST>(t1, sc);
  const float ibw = M::rcp(bw), ibh = M::rcp(bh);
  const float s = bh * ibw;
  const float st = d1 + d0 - 2.0f * s;
  const float l2s = 2.0f * M::log(s);
  rqs_bin_eval<INV, FAST>(v, x0, y0, bw, bh, ibw, ibh, s, st, d0, d1, l2s, out, ld);
  if (v <= sc.lo) {          // bin 0 was selected: d0 = slope[0]
    out = INV ? M::div(v - sc.lo, d0) + sc.lo : fmaf(v - sc.lo, d0, sc.lo);
    ld = INV ? -M::log(d0) : M::log(d0);
  }
  if (v >= sc.hi) {          // bin K-1 was selected: d1 = slope[K]
    out = INV ? M::div(v - sc.hi, d1) + sc.hi : fmaf(v - sc.hi, d1, sc.hi);
    ld = INV ? -M::log(d1) : M::log(d1);
  }
}

// ---------------------------------------------------------------------------
// Conditioner MLP (flows.py:57-84): [c, v_0..v_{d-1}] -> H (relu) -> ... ->
// H (relu) -> P.  One sample per lane; `w` is a wave-uniform pointer into the
// prepared weights, so every weight is a scalar (SGPR) operand of the
// per-lane FMA: no LDS or VGPR traffic for weights at all.  The d inputs v_q
// are read from this thread's own LDS column `col[q * stride]`.
// ---------------------------------------------------------------------------
// acc[j] += sum_i in[i] * W[i][j] for a wave-uniform row-major W[R][N].
// The rows are consumed in groups of G: each group is G*N scalar loads
// (s_load_dwordx16) followed by G*N per-lane FMAs with an SGPR operand.  The
// scheduling barrier after every group stops hipcc from hoisting ALL of the
// layer's scalar loads to the top, which needs ~600 SGPRs and spills them to
// VGPR lanes (v_writelane/v_readlane: 940 extra VALU ops per conditioner in the
// first build).  Scalar-load latency is hidden by the other waves of the SIMD.
template <int R, int N, int G>
__device__ __forceinline__ void dense_acc(uniform_ptr W, const float (&in)[R], float (&acc)[N]) {
  static_assert(R % G == 0, "row group must divide the row count");
#pragma unroll
  for (int g = 0; g < R / G; ++g) {
    float wv[G * N];
#pragma unroll
    for (int t = 0; t < G * N; ++t) wv[t] = W[g * G * N + t];
#pragma unroll
    for (int r = 0; r < G; ++r) {
#pragma unroll
      for (int j = 0; j < N; ++j) acc[j] = fmaf(in[g * G + r], wv[r * N + j], acc[j]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Pins every element in a VGPR at this point of the program: without it the
// compiler sinks whole accumulation chains (and the 16x16 weights they need,
// as spilled SGPRs) down to their first use in the spline code.
template <int N>
__device__ __forceinline__ void materialize(float (&v)[N]) {
#pragma unroll
  for (int j = 0; j < N; ++j) asm volatile("" : "+v"(v[j]));
}

template <int N>
__device__ __forceinline__ void load_row(uniform_ptr p, float (&v)[N]) {
#pragma unroll
  for (int j = 0; j < N; ++j) v[j] = p[j];
}

constexpr int row_group(int R, int N) { return (R % 2 == 0 && N <= 16) ? 2 : 1; }

template <int H, int P>
__device__ __forceinline__ void conditioner(uniform_ptr w, int d, int M, float c,
                                            const float* col, int first_idx, int idx_step,
                                            int stride, float (&th)[P]) {
  float h[H];
  w = launder(w);
  uniform_ptr b0 = w + (1 + d) * H;
  {
    float wc[H], bb[H];
    load_row<H>(w, wc);
    load_row<H>(b0, bb);
#pragma unroll
    for (int j = 0; j < H; ++j) h[j] = fmaf(c, wc[j], bb[j]);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int q = 0; q < d; ++q) {              // runtime loop: d is not a template arg
    const float v = col[(first_idx + q * idx_step) * stride];
    float wr[H];
    load_row<H>(w + (1 + q) * H, wr);
#pragma unroll
    for (int j = 0; j < H; ++j) h[j] = fmaf(v, wr[j], h[j]);
  }
#pragma unroll
  for (int j = 0; j < H; ++j) h[j] = fmaxf(h[j], 0.0f);
  materialize<H>(h);
  w = b0 + H;
  for (int m = 1; m < M; ++m) {
    float g[H];
    uniform_ptr b = w + H * H;
    load_row<H>(b, g);
    __builtin_amdgcn_sched_barrier(0);
    dense_acc<H, H, row_group(H, H)>(w, h, g);
#pragma unroll
    for (int j = 0; j < H; ++j) h[j] = fmaxf(g[j], 0.0f);
    materialize<H>(h);
    w = b + H;
  }
  load_row<P>(w + H * P, th);
  __builtin_amdgcn_sched_barrier(0);
  dense_acc<H, P, row_group(H, P)>(w, h, th);
  materialize<P>(th);
}

__host__ __device__ inline int64_t cond_floats(int d, int H, int M, int P) {
  return (int64_t)(1 + d) * H + H + (int64_t)(M - 1) * (H * H + H) + (int64_t)H * P + P;
}

// ---------------------------------------------------------------------------
// Philox4x32-10 (same stream as oracle/cnf_oracle.c).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

}  // namespace cnf
