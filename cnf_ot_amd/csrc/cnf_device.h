// cnf_device.h -- device-side building blocks of the conditional RQS flow for
// gfx950 (CDNA4): prepared `first` table layout, spline evaluation from the
// table (LDS gather) and from per-lane conditioner output (registers), and the
// time-conditioned MLP conditioner with wave-uniform weights.
//
// Everything is templated on the per-lane value type T:
//   T = float : one sample per lane;
//   T = v2f   : two samples per lane in a 64-bit register pair, so the bulk of
//               the arithmetic issues as packed fp32 (v_pk_fma_f32 /
//               v_pk_mul_f32 / v_pk_add_f32).  A wave64 VALU instruction
//               occupies the SIMD for 4 cycles on gfx950 (measured:
//               profiles/r01a, SQ_ACTIVE_INST_VALU = 100 % at 4.1 cycles per
//               instruction), so only packed math reaches the fp32 peak.
//
// Reference behaviour restated (never copied; the reference is Python/JAX):
//   spline       distrax.RationalQuadraticSpline, call site flows.py:124-132
//                (algorithm: SURVEY.md Appendix A)
//   conditioner  cnf_ot/models/flows.py:46-86
//   coupling     cnf_ot/models/autoregressive.py:76-136
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace cnf {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));

// Pointer into the AMDGPU constant address space (4).  Global and constant
// addresses coincide; a load through this type from a wave-uniform address is
// always selected as a scalar-cache load (s_load_dword*), so the value lands
// in SGPRs and feeds the per-lane FMAs as a scalar operand.  (Through a plain
// global pointer hipcc falls back to per-lane global_load as soon as the
// kernel also stores to global memory, because it cannot prove the weights
// are not clobbered.)  The prepared buffer is never written by a flow kernel.
typedef const float __attribute__((address_space(4)))* uniform_ptr;

__device__ __forceinline__ uniform_ptr as_uniform(const float* p) { return (uniform_ptr)p; }

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
//   [F * KP + k]   field F of bin k of the `first` spline, F in TabField,
//                  KP = tab_stride(K) (structure-of-arrays: a lane gathers each
//                  field with one ds_read_b32 straight into the register it is
//                  used from)
//   [hdr_floats(K), ...)   conditioner weights, flat layout of the C ABI
// The `first` spline is shared by every layer and ignores c (flows.py:47-55,
// autoregressive.py:88-92), so it is normalised ONCE per parameter set, in
// float64, instead of once per sample per layer.
// ---------------------------------------------------------------------------
enum TabField {
  F_X0 = 0, F_Y0, F_BW, F_BH, F_IBW, F_IBH, F_S, F_ST, F_D0, F_D1, F_L2S,
  F_XK,      // x knots [K+1]
  F_YK,      // y knots [K+1]
  F_TAIL,    // d_lo, d_hi, log d_lo, log d_hi, 1/d_lo, 1/d_hi
  F_XKB,     // -2^60 x x knot j, each stored TWICE ([2 j], [2 j + 1]): the ds_read_b64 operand of the packed
  F_XKB2,    //   clamp-FMA that forms the bin masks of a sample pair (bin_of_pairs); two fields wide
  F_YKB,     // the same for the y knots
  F_YKB2,
  F_COUNT
};
enum { T_DLO = 0, T_DHI, T_LOG_DLO, T_LOG_DHI, T_INV_DLO, T_INV_DHI };

__host__ __device__ constexpr int tab_stride(int K) { return K + 1 < 6 ? 6 : K + 1; }
__host__ __device__ constexpr int tab_off(int field, int K) { return field * tab_stride(K); }
__host__ __device__ constexpr int hdr_floats(int K) { return (F_COUNT * tab_stride(K) + 3) & ~3; }

template <class R> struct SplineConstsT {
  R lo, hi;        // range_min, range_max
  R span_eff;      // (hi - lo) - K * min_bin
  R min_bin;
  R min_slope;
  R sp_offset;     // log(exp(1 - min_slope) - 1)
};
typedef SplineConstsT<float> SplineConsts;

// The constants of a kernel argument as individually pinned scalars.  Read through a reference into the (4-byte
// aligned) argument struct, the vectoriser pairs lo / hi / span_eff into <2 x float> loads and routes them through
// a 16-byte scratch copy that every tile then re-loads with vector-memory latency (seen in flow_pwl_kernel).
__device__ __forceinline__ SplineConsts sc_scalars(const SplineConsts& s) {
  float lo = s.lo, hi = s.hi, span = s.span_eff, mb = s.min_bin, ms = s.min_slope, so = s.sp_offset;
  asm volatile("" : "+s"(lo), "+s"(hi), "+s"(span), "+s"(mb), "+s"(ms), "+s"(so));
  SplineConsts r;
  r.lo = lo; r.hi = hi; r.span_eff = span; r.min_bin = mb; r.min_slope = ms; r.sp_offset = so;
  return r;
}

// ---------------------------------------------------------------------------
// Elementwise helpers over T in {float, v2f}
// ---------------------------------------------------------------------------
// T = double is the reference's own dtype (jax_enable_x64, solvers.py:23): one
// sample per lane, ocml math, float64 table / constants / IO.  It exists for
// exact-mode parity (1e-12 vs the oracle), not for speed.
template <class T> struct Lanes;
template <> struct Lanes<float> { static constexpr int N = 1; typedef int index; typedef float real; };
template <> struct Lanes<v2f> { static constexpr int N = 2; typedef v2i index; typedef float real; };
template <> struct Lanes<double> { static constexpr int N = 1; typedef int index; typedef double real; };

template <class T, class A> __device__ __forceinline__ T splat(A a) {
  if constexpr (std::is_same<T, v2f>::value) return v2f{(float)a, (float)a};
  else return (T)a;
}

__device__ __forceinline__ double vfma(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ double vfma(float a, double b, double c) { return fma((double)a, b, c); }
__device__ __forceinline__ double vmax(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ double vabs(double a) { return fabs(a); }
__device__ __forceinline__ double vrelu(double a) { return fmax(a, 0.0); }
__device__ __forceinline__ double clip01(double z) { return fmin(fmax(z, 0.0), 1.0); }
__device__ __forceinline__ double vclamp(double z, double hi) { return fmin(fmax(z, 0.0), hi); }
__device__ __forceinline__ bool vge(double a, double b) { return a >= b; }
__device__ __forceinline__ bool vle(double a, double b) { return a <= b; }
__device__ __forceinline__ bool vlt(double a, double b) { return a < b; }
__device__ __forceinline__ double vsel(bool m, double a, double b) { return m ? a : b; }

__device__ __forceinline__ float vfma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ v2f vfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f vfma(float a, v2f b, v2f c) { return __builtin_elementwise_fma(v2f{a, a}, b, c); }
__device__ __forceinline__ v2f vfma(v2f a, float b, v2f c) { return __builtin_elementwise_fma(a, v2f{b, b}, c); }
__device__ __forceinline__ v2f vfma(v2f a, v2f b, float c) { return __builtin_elementwise_fma(a, b, v2f{c, c}); }

__device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ v2f vmax(v2f a, v2f b) { return v2f{fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
__device__ __forceinline__ float vabs(float a) { return fabsf(a); }
__device__ __forceinline__ v2f vabs(v2f a) { return v2f{fabsf(a.x), fabsf(a.y)}; }
__device__ __forceinline__ float vrelu(float a) { return fmaxf(a, 0.0f); }
__device__ __forceinline__ v2f vrelu(v2f a) { return v2f{fmaxf(a.x, 0.0f), fmaxf(a.y, 0.0f)}; }
// clamp to [0, 1] / [0, hi] (hi > 0) in one v_med3_f32
__device__ __forceinline__ float clip01(float z) { return __builtin_amdgcn_fmed3f(z, 0.0f, 1.0f); }
__device__ __forceinline__ v2f clip01(v2f z) { return v2f{clip01(z.x), clip01(z.y)}; }
__device__ __forceinline__ float vclamp(float z, float hi) { return __builtin_amdgcn_fmed3f(z, 0.0f, hi); }
__device__ __forceinline__ v2f vclamp(v2f z, v2f hi) { return v2f{vclamp(z.x, hi.x), vclamp(z.y, hi.y)}; }

__device__ __forceinline__ bool vge(float a, float b) { return a >= b; }
__device__ __forceinline__ v2i vge(v2f a, v2f b) { return a >= b; }
__device__ __forceinline__ v2i vge(v2f a, float b) { return a >= v2f{b, b}; }
__device__ __forceinline__ bool vle(float a, float b) { return a <= b; }
__device__ __forceinline__ v2i vle(v2f a, float b) { return a <= v2f{b, b}; }
__device__ __forceinline__ bool vlt(float a, float b) { return a < b; }
__device__ __forceinline__ v2i vlt(v2f a, float b) { return a < v2f{b, b}; }

__device__ __forceinline__ float vsel(bool m, float a, float b) { return m ? a : b; }
__device__ __forceinline__ v2f vsel(v2i m, v2f a, v2f b) { return v2f{m.x ? a.x : b.x, m.y ? a.y : b.y}; }
__device__ __forceinline__ bool vany(bool m) { return m; }
__device__ __forceinline__ bool vany(v2i m) { return (m.x | m.y) != 0; }

// "Might this lane need the linear tails?" -- a cheap, slightly conservative
// test (the tail code itself applies the exact v <= lo / v >= hi masks): three
// VALU instructions for a sample pair instead of four compares and the mask
// arithmetic.
template <class R> __device__ __forceinline__ bool maybe_outside(R v, R lo, R hi) { return v <= lo || v >= hi; }
__device__ __forceinline__ bool maybe_outside(v2f v, float lo, float hi) {
  const float mid = 0.5f * (lo + hi), half = 0.5f * (hi - lo) * 0.99999f;
  const v2f d = v - mid;
  return !(fmaxf(fabsf(d.x), fabsf(d.y)) < half);     // NaN counts as outside
}

// ---------------------------------------------------------------------------
// Math policy.  FAST=false: ocml expf/logf/sqrtf and IEEE division.
// FAST=true: hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32 /
// v_sqrt_f32, 1 ulp each); the scalings around them are packed for v2f.
// exp(x) = 2^(x*log2 e): the product's rounding costs |x| * 6e-8 relative,
// which only matters for terms that are negligible anyway (softmax terms far
// below the maximum, softplus arguments far from 0).
// ---------------------------------------------------------------------------
template <bool FAST> struct Math;

template <> struct Math<false> {
  static __device__ __forceinline__ float exp(float x) { return expf(x); }
  static __device__ __forceinline__ float log(float x) { return logf(x); }
  static __device__ __forceinline__ float rcp(float x) { return 1.0f / x; }
  static __device__ __forceinline__ float div(float a, float b) { return a / b; }
  static __device__ __forceinline__ float sqrt(float x) { return sqrtf(x); }
  static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
  static __device__ __forceinline__ double log(double x) { return ::log(x); }
  static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
  static __device__ __forceinline__ double div(double a, double b) { return a / b; }
  static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
  static __device__ __forceinline__ v2f exp(v2f x) { return v2f{expf(x.x), expf(x.y)}; }
  static __device__ __forceinline__ v2f log(v2f x) { return v2f{logf(x.x), logf(x.y)}; }
  static __device__ __forceinline__ v2f rcp(v2f x) { return v2f{1.0f / x.x, 1.0f / x.y}; }
  static __device__ __forceinline__ v2f div(v2f a, v2f b) { return v2f{a.x / b.x, a.y / b.y}; }
  static __device__ __forceinline__ v2f sqrt(v2f x) { return v2f{sqrtf(x.x), sqrtf(x.y)}; }
};

constexpr float LOG2E = 1.44269502162933349609375f;
constexpr float LN2 = 0.693147182464599609375f;

template <> struct Math<true> {
  static __device__ __forceinline__ float exp(float x) { return __builtin_amdgcn_exp2f(x * LOG2E); }
  static __device__ __forceinline__ float log(float x) { return __builtin_amdgcn_logf(x) * LN2; }
  static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
  static __device__ __forceinline__ float div(float a, float b) {
    const float r = __builtin_amdgcn_rcpf(b), q = a * r;
    return fmaf(fmaf(-b, q, a), r, q);                   // one Newton step
  }
  static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
  static __device__ __forceinline__ v2f exp(v2f x) {
    const v2f t = x * LOG2E;
    return v2f{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
  }
  static __device__ __forceinline__ float exp2(float x) { return __builtin_amdgcn_exp2f(x); }
  static __device__ __forceinline__ v2f exp2(v2f x) { return v2f{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }
  static __device__ __forceinline__ v2f log(v2f x) {
    return v2f{__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y)} * LN2;
  }
  static __device__ __forceinline__ v2f rcp(v2f x) {
    return v2f{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
  }
  static __device__ __forceinline__ v2f div(v2f a, v2f b) {
    const v2f r = rcp(b), q = a * r;
    return vfma(vfma(-b, q, a), r, q);
  }
  static __device__ __forceinline__ v2f sqrt(v2f x) {
    return v2f{__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)};
  }
};

// softplus(t + offset) + m  (distrax _normalize_knot_slopes)
template <bool FAST, class T, bool OFFSET_ADDED = false>
__device__ __forceinline__ T knot_slope(T t, const SplineConstsT<typename Lanes<T>::real> sc) {
  const T v = OFFSET_ADDED ? t : t + sc.sp_offset;
  const T av = vabs(v);
  const T e = Math<FAST>::exp(-av);
  // log1p(e), e in (0,1].  Both forms are evaluated and selected: a branch
  // here would split the wave's straight-line code.
  typedef typename Lanes<T>::real R;
  // float64: log1p proper (the oracle / distrax use softplus = logaddexp)
  if constexpr (std::is_same<T, double>::value) {
    return vrelu(v) + ::log1p(e) + sc.min_slope;
  } else {
    const T l_log = Math<FAST>::log(e + 1.0f);
    const T l_ser = vfma(e * -0.5f, e, e);
    const T l = vsel(vlt(e, (R)1e-4f), l_ser, l_log);
    // relu(v) = (v + |v|) / 2: two packed ops for a sample pair
    return vfma(v + av, splat<T>(0.5f), l) + sc.min_slope;
  }
}

// Shared tail of both directions: from the selected bin to (out, logdet).
// INV=false: distrax _rational_quadratic_spline_fwd; INV=true: ..._inv.
// The inverse solves a z^2 + b z + c = 0 (distrax: a = s - b, b = d0 - st w,
// c = -s w with w = (v - y0) / bh) scaled through by bh -- the root does not
// change and 1 / bh is never needed.  L2S_GIVEN: 2 log s comes from the caller
// (the prepared table); otherwise it is folded into the one logarithm.
// ARG: `ld` receives the ARGUMENT of that logarithm (the spline's derivative f'), not +-log of it: a caller that
// evaluates several splines multiplies the arguments and takes one logarithm.
template <bool INV, bool FAST, class T, bool L2S_GIVEN = true, bool ARG = false>
__device__ __forceinline__ void rqs_bin_eval(T v, T x0, T y0, T bw, T bh, T ibw, T s, T st,
                                             T d0, T d1, T l2s, T& out, T& ld) {
  using M = Math<FAST>;
  T z;
  if (INV) {
    const T dy = vclamp(v - y0, bh);              // bh * w, w clipped to [0, 1]
    const T c = -s * dy;
    const T b = vfma(-st, dy, d0 * bh);
    const T a = vfma(s, bh, -b);
    const T disc = vfma(b, b, a * c * -4.0f);
    z = clip01(M::div(c * -2.0f, b + M::sqrt(disc)));
    out = vfma(bw, z, x0);
  } else {
    z = clip01((v - x0) * ibw);
  }
  const T sq_z = z * z;
  const T z1mz = z - sq_z;
  const T omz = 1.0f - z;
  const T den = vfma(st, z1mz, s);
  const T iden = M::rcp(den);
  if (!INV) out = vfma(bh * vfma(s, sq_z, d0 * z1mz), iden, y0);
  const T num2 = vfma(d1, sq_z, vfma(s * 2.0f, z1mz, d0 * omz * omz));
  // 2 log s + log(num2) - 2 log(den)
  if constexpr (ARG) {
    static_assert(!L2S_GIVEN, "the argument form folds s^2 in");
    const T q = s * iden;
    ld = num2 * q * q;
  } else {
    T ldf;
    if (L2S_GIVEN) {
      ldf = l2s + M::log(num2 * iden * iden);
    } else {
      const T q = s * iden;
      ldf = M::log(num2 * q * q);
    }
    ld = INV ? -ldf : ldf;
  }
}

// ---------------------------------------------------------------------------
// Spline of the shared `first` parameters: per-lane bin index, LDS gather of
// the bin's pre-normalised constants.  `tab` is the prepared header in LDS.
// ---------------------------------------------------------------------------
template <int K, class R> __device__ __forceinline__ int bin_of(const R* pos, R v) {
  int k = 0;
#pragma unroll
  for (int j = 1; j < K; ++j) k += (v >= pos[j]) ? 1 : 0;
  return k;
}
// Two samples: 0/1 masks [v > knot_j] from ONE packed FMA with the clamp modifier per knot,
// clamp(v 2^60 - knot_j 2^60) (a single rounding: the sign is that of v - knot_j), summed in fp32 and converted
// once (v exactly on a knot lands in the lower bin; the spline is continuous there).  The prepared table holds
// -2^60 knot_j twice in a row (F_XKB / F_YKB), so the FMA's third operand is one ds_read_b64 from a uniform
// address: per knot one LDS read and one VALU instruction (11 VALU per pair in all, against 26 for per-sample
// sign-bit arithmetic).
template <int K> __device__ __forceinline__ v2i bin_of_pairs(const float* kb, v2f v) {
  const v2f big = v2f{1.152921504606846976e18f, 1.152921504606846976e18f};       // 2^60
  v2f c = v2f{0.0f, 0.0f};
#pragma unroll
  for (int j = 1; j < K; ++j) {
    const v2f nk = *reinterpret_cast<const v2f*>(kb + 2 * j);
    v2f m;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(m) : "v"(v), "v"(big), "v"(nk));
    c = j == 1 ? m : c + m;
  }
  return v2i{(int)c.x, (int)c.y};
}
template <int K, class R> __device__ __forceinline__ R gather(const R* tab, int f, int k) {
  return tab[tab_off(f, K) + k];
}
template <int K> __device__ __forceinline__ v2f gather(const float* tab, int f, v2i k) {
  return v2f{tab[tab_off(f, K) + k.x], tab[tab_off(f, K) + k.y]};
}

// The selected bin's row of the prepared table.  For sample pairs: two LDS
// pointers (table base + bin, one v_add each); every field is then ONE
// ds_read_b32 per sample with the field offset as the instruction's immediate,
// landing directly in its half of the pair.  volatile keeps the reads from being
// merged into ds_read2_b32, whose results would need a v_mov per field to re-pair.
template <class T> struct BinRow {
  const typename Lanes<T>::real* p;
  __device__ __forceinline__ BinRow(const typename Lanes<T>::real* tab, int k) : p(tab + k) {}
  template <int K> __device__ __forceinline__ T get(int f) const { return p[tab_off(f, K)]; }
};
template <> struct BinRow<v2f> {
  typedef const volatile float __attribute__((address_space(3))) * lds_ptr;
  lds_ptr px, py;
  __device__ __forceinline__ BinRow(const float* tab, v2i k) {
    lds_ptr t = (lds_ptr)(uintptr_t)(uint32_t)(uintptr_t)tab;       // low 32 bits of a flat LDS address = LDS offset
    px = t + k.x;
    py = t + k.y;
  }
  template <int K> __device__ __forceinline__ v2f get(int f) const { return v2f{px[tab_off(f, K)], py[tab_off(f, K)]}; }
};

template <int K, bool INV, bool FAST, class T, bool ARG = false>
__device__ __forceinline__ void table_spline(const typename Lanes<T>::real* tab, T v,
                                             const SplineConstsT<typename Lanes<T>::real> sc, T& out, T& ld) {
  typedef typename Lanes<T>::real R;
  typename Lanes<T>::index k;
  if constexpr (std::is_same<T, v2f>::value) k = bin_of_pairs<K>(tab + tab_off(INV ? F_YKB : F_XKB, K), v);
  else k = bin_of<K>(tab + tab_off(INV ? F_YK : F_XK, K), v);
  const BinRow<T> row(tab, k);
  // (the pair rows are read through volatile pointers: a field the direction does not use must not be asked for)
  const T bw = INV ? row.template get<K>(F_BW) : splat<T>(0.0f);          // the inverse never divides by bw ...
  const T ibw = INV ? splat<T>(0.0f) : row.template get<K>(F_IBW);        // ... the forward map only does
  if constexpr (ARG)
    rqs_bin_eval<INV, FAST, T, false, true>(v, row.template get<K>(F_X0), row.template get<K>(F_Y0), bw,
                                            row.template get<K>(F_BH), ibw,
                                            row.template get<K>(F_S), row.template get<K>(F_ST), row.template get<K>(F_D0),
                                            row.template get<K>(F_D1), splat<T>(0.0f), out, ld);
  else
    rqs_bin_eval<INV, FAST, T>(v, row.template get<K>(F_X0), row.template get<K>(F_Y0), bw,
                               row.template get<K>(F_BH), ibw,
                               row.template get<K>(F_S), row.template get<K>(F_ST), row.template get<K>(F_D0),
                               row.template get<K>(F_D1), row.template get<K>(F_L2S), out, ld);
  if (maybe_outside(v, sc.lo, sc.hi)) {   // linear tails (rare: |v| >= 10)
    const auto below = vle(v, sc.lo);
    const auto above = vge(v, sc.hi);
    const R* tl = tab + tab_off(F_TAIL, K);
    const T lo_out = INV ? vfma(v - sc.lo, splat<T>(tl[T_INV_DLO]), splat<T>(sc.lo))
                         : vfma(v - sc.lo, splat<T>(tl[T_DLO]), splat<T>(sc.lo));
    const T hi_out = INV ? vfma(v - sc.hi, splat<T>(tl[T_INV_DHI]), splat<T>(sc.hi))
                         : vfma(v - sc.hi, splat<T>(tl[T_DHI]), splat<T>(sc.hi));
    out = vsel(below, lo_out, out);
    ld = vsel(below, ARG ? splat<T>(tl[T_DLO]) : splat<T>(INV ? -tl[T_LOG_DLO] : tl[T_LOG_DLO]), ld);
    out = vsel(above, hi_out, out);
    ld = vsel(above, ARG ? splat<T>(tl[T_DHI]) : splat<T>(INV ? -tl[T_LOG_DHI] : tl[T_LOG_DHI]), ld);
  }
}

// ---------------------------------------------------------------------------
// The same spline for sample pairs fed from the conditioner tables (cnf_pwl.h).
// ---------------------------------------------------------------------------
// 0/1 mask of (d > 0) for a sample pair: clamp(d * 2^60) -- one v_pk_mul_f32
// with the clamp modifier (differences below 2^-60 count as ties, i.e. the
// lower bin, where the spline is continuous anyway).
__device__ __forceinline__ v2f step_mask(v2f d, v2f big) {
  v2f m;
  asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(m) : "v"(d), "v"(big));
  return m;
}

// cond_spline for sample pairs fed from the piecewise-linear tables.  Rows hold the softmax logits in log2 units,
// shifted per piece so that each group's maximum is ~0 near the samples the piece serves, and the slope logits
// with the softplus offset added, also in log2 units (cnf_pwl.h).  The bin is selected arithmetically with 0/1 masks m_k = [v > knot_k]
// in packed FMAs (40 packed instructions per pair instead of 8 compares + 48 v_cndmask):
//   x0 = lo + sum m_k w_(k-1)  (bitwise the running knot),
//   bw, bh, t0, t1 = sum o_k (w_k, h_k, t_k, t_(k+1))  with the one-hot o_k = m_k - m_(k+1)
// -- products with 0 or 1 and sums of zeros: bitwise the selected values.
// SHIFT_FREE (the table builder has checked the sample's cell: each group's maximum within +-4 of zero, slope
// logits in [-3, 40]): e_k = 2^th_k directly -- no running maximum, no subtraction; ONE reciprocal for both
// softmax sums; slopes as ln2 log2(1 + 2^t) with no small-argument series.
// Otherwise the general form: running maximum, the log1p series for slope arguments near zero.
template <int K, bool INV, bool FAST, bool SHIFT_FREE>
__device__ __forceinline__ void cond_spline_masked(const v2f (&th)[3 * K + 1], v2f v,
                                                   const SplineConsts sc, v2f& out, v2f& ld) {
  using M = Math<FAST>;
  typedef v2f T;
  T ew[K], eh[K];
  T aw, ah;
  if constexpr (SHIFT_FREE) {
#pragma unroll
    for (int k = 0; k < K; ++k) { ew[k] = M::exp2(th[k]); eh[k] = M::exp2(th[K + k]); }
    T sw = ew[0] + ew[1], sh = eh[0] + eh[1];
#pragma unroll
    for (int k = 2; k < K; ++k) { sw += ew[k]; sh += eh[k]; }
    const T r = M::rcp(sw * sh) * sc.span_eff;
    aw = r * sh; ah = r * sw;
  } else {
    T mw = th[0], mh = th[K];
#pragma unroll
    for (int k = 1; k < K; ++k) { mw = vmax(mw, th[k]); mh = vmax(mh, th[K + k]); }
#pragma unroll
    for (int k = 0; k < K; ++k) { ew[k] = M::exp2(th[k] - mw); eh[k] = M::exp2(th[K + k] - mh); }
    T sw = ew[0], sh = eh[0];
#pragma unroll
    for (int k = 1; k < K; ++k) { sw += ew[k]; sh += eh[k]; }
    aw = M::rcp(sw) * sc.span_eff; ah = M::rcp(sh) * sc.span_eff;
  }
  const T big = splat<T>(1.152921504606846976e18f);       // 2^60
  T px = splat<T>(sc.lo), py = splat<T>(sc.lo);            // running knot k
  T wp = vfma(ew[0], aw, splat<T>(sc.min_bin)), hp = vfma(eh[0], ah, splat<T>(sc.min_bin));   // bin k-1
  T x0 = px, y0 = py;
  T mprev = splat<T>(1.0f), bw = splat<T>(0.0f), bh = splat<T>(0.0f), t0 = splat<T>(0.0f), t1 = splat<T>(0.0f);
#pragma unroll
  for (int k = 1; k < K; ++k) {
    px += wp;
    py += hp;
    const T m = step_mask(v - (INV ? py : px), big);
    const T o = mprev - m;                                  // one-hot of bin k-1
    bw = k == 1 ? o * wp : vfma(o, wp, bw);
    bh = k == 1 ? o * hp : vfma(o, hp, bh);
    t0 = k == 1 ? o * th[2 * K] : vfma(o, th[2 * K + k - 1], t0);
    t1 = k == 1 ? o * th[2 * K + 1] : vfma(o, th[2 * K + k], t1);
    x0 = vfma(m, wp, x0);
    y0 = vfma(m, hp, y0);
    if (k == K - 1) { wp = sc.hi - px; hp = sc.hi - py; }   // last knot is exactly hi
    else { wp = vfma(ew[k], aw, splat<T>(sc.min_bin)); hp = vfma(eh[k], ah, splat<T>(sc.min_bin)); }
    mprev = m;
  }
  bw = vfma(mprev, wp, bw);
  bh = vfma(mprev, hp, bh);
  t0 = vfma(mprev, th[3 * K - 1], t0);
  t1 = vfma(mprev, th[3 * K], t1);
  T d0, d1;
  if constexpr (SHIFT_FREE) {      // softplus(t) + m = ln2 log2(1 + 2^(t log2 e)) + m, t in [-3, 40]
    const T l0 = v2f{__builtin_amdgcn_logf(M::exp2(t0).x + 1.0f), __builtin_amdgcn_logf(M::exp2(t0).y + 1.0f)};
    const T l1 = v2f{__builtin_amdgcn_logf(M::exp2(t1).x + 1.0f), __builtin_amdgcn_logf(M::exp2(t1).y + 1.0f)};
    d0 = vfma(l0, splat<T>(LN2), splat<T>(sc.min_slope));
    d1 = vfma(l1, splat<T>(LN2), splat<T>(sc.min_slope));
  } else {
    // natural units.  The product must be ROUNDED before knot_slope forms v + |v|: contracted into an FMA,
    // t LN2 + |round(t LN2)| is the product's rounding error instead of 0, and a slope at its floor of 1e-4 is
    // then off by up to ulp(|t|) -- found by the soak (scripts/soak_pwl.py), tails with |u| >> 16.
    T n0 = t0 * LN2, n1 = t1 * LN2;
    asm volatile("" : "+v"(n0), "+v"(n1));
    d0 = knot_slope<FAST, T, true>(n0, sc);
    d1 = knot_slope<FAST, T, true>(n1, sc);
  }
  const T ibw = M::rcp(bw);
  const T s = bh * ibw;
  const T st = d1 + d0 - s * 2.0f;
  rqs_bin_eval<INV, FAST, T, false>(v, x0, y0, bw, bh, ibw, s, st, d0, d1, s, out, ld);
  if (maybe_outside(v, sc.lo, sc.hi)) {
    const auto below = vle(v, sc.lo);          // bin 0 was selected: d0 = slope[0]
    const auto above = vge(v, sc.hi);          // bin K-1 was selected: d1 = slope[K]
    const T lo_out = INV ? M::div(v - sc.lo, d0) + sc.lo : vfma(v - sc.lo, d0, splat<T>(sc.lo));
    const T hi_out = INV ? M::div(v - sc.hi, d1) + sc.hi : vfma(v - sc.hi, d1, splat<T>(sc.hi));
    const T ld0 = M::log(d0), ld1 = M::log(d1);
    out = vsel(below, lo_out, out);
    ld = vsel(below, INV ? -ld0 : ld0, ld);
    out = vsel(above, hi_out, out);
    ld = vsel(above, INV ? -ld1 : ld1, ld);
  }
}

// cond_spline_masked with the table rows evaluated where they are cheapest (round 2).  The two samples of a lane sit
// in unrelated rows, so a row's 16 FMAs cannot be packed ACROSS the samples -- but neighbouring parameters of ONE
// sample are neighbours in its row: qa / qb hold the 2K softmax logits of sample a / b as K pairs
// (logit 2j, logit 2j+1), one v_pk_fma_f32 per pair (K per sample instead of 2K v_fma_f32), and the exponentials
// -- scalar instructions anyway -- write their results into the sample-pair layout the rest of the spline uses.
// The slope logits are not evaluated for all K + 1 knots and then selected with masks: the bin index (the sum of
// the 0/1 masks) addresses the row a second time and `slopes(ka, kb, ta, tb)` returns (t_k, t_k+1) of each sample
// from one packed FMA.  Same values bit for bit as cond_spline_masked on the same rows.
template <int K, bool INV, bool FAST, bool SHIFT_FREE, bool ARG = false, class SlopeFetch>
__device__ __forceinline__ void cond_spline_rows(const v2f (&qa)[K], const v2f (&qb)[K], SlopeFetch&& slopes, v2f v,
                                                 const SplineConsts sc, v2f& out, v2f& ld) {
  using M = Math<FAST>;
  typedef v2f T;
  auto la = [&](int j) { return qa[j >> 1][j & 1]; };
  auto lb = [&](int j) { return qb[j >> 1][j & 1]; };
  T ew[K], eh[K];
  T aw, ah;
  if constexpr (SHIFT_FREE) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      ew[k] = v2f{__builtin_amdgcn_exp2f(la(k)), __builtin_amdgcn_exp2f(lb(k))};
      eh[k] = v2f{__builtin_amdgcn_exp2f(la(K + k)), __builtin_amdgcn_exp2f(lb(K + k))};
    }
    T sw = ew[0] + ew[1], sh = eh[0] + eh[1];
#pragma unroll
    for (int k = 2; k < K; ++k) { sw += ew[k]; sh += eh[k]; }
    const T r = M::rcp(sw * sh) * sc.span_eff;
    aw = r * sh; ah = r * sw;
  } else {
    float mwa = la(0), mwb = lb(0), mha = la(K), mhb = lb(K);
#pragma unroll
    for (int k = 1; k < K; ++k) {
      mwa = fmaxf(mwa, la(k)); mwb = fmaxf(mwb, lb(k));
      mha = fmaxf(mha, la(K + k)); mhb = fmaxf(mhb, lb(K + k));
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      ew[k] = v2f{__builtin_amdgcn_exp2f(la(k) - mwa), __builtin_amdgcn_exp2f(lb(k) - mwb)};
      eh[k] = v2f{__builtin_amdgcn_exp2f(la(K + k) - mha), __builtin_amdgcn_exp2f(lb(K + k) - mhb)};
    }
    T sw = ew[0], sh = eh[0];
#pragma unroll
    for (int k = 1; k < K; ++k) { sw += ew[k]; sh += eh[k]; }
    aw = M::rcp(sw) * sc.span_eff; ah = M::rcp(sh) * sc.span_eff;
  }
  const T big = splat<T>(1.152921504606846976e18f);       // 2^60
  T px = splat<T>(sc.lo), py = splat<T>(sc.lo);            // running knot k
  T wp = vfma(ew[0], aw, splat<T>(sc.min_bin)), hp = vfma(eh[0], ah, splat<T>(sc.min_bin));   // bin k-1
  T x0 = px, y0 = py;
  T mprev = splat<T>(1.0f), bw = splat<T>(0.0f), bh = splat<T>(0.0f), kf = splat<T>(0.0f);
#pragma unroll
  for (int k = 1; k < K; ++k) {
    px += wp;
    py += hp;
    const T m = step_mask(v - (INV ? py : px), big);
    const T o = mprev - m;                                  // one-hot of bin k-1
    bw = k == 1 ? o * wp : vfma(o, wp, bw);
    bh = k == 1 ? o * hp : vfma(o, hp, bh);
    kf = k == 1 ? m : kf + m;                               // masks are monotone: their sum is the bin index
    x0 = vfma(m, wp, x0);
    y0 = vfma(m, hp, y0);
    if (k == K - 1) { wp = sc.hi - px; hp = sc.hi - py; }   // last knot is exactly hi
    else { wp = vfma(ew[k], aw, splat<T>(sc.min_bin)); hp = vfma(eh[k], ah, splat<T>(sc.min_bin)); }
    mprev = m;
  }
  bw = vfma(mprev, wp, bw);
  bh = vfma(mprev, hp, bh);
  v2f ta, tb;                                               // (t_k, t_k+1) of sample a, of sample b
  slopes((int)kf.x, (int)kf.y, ta, tb);
  T d0, d1;
  if constexpr (SHIFT_FREE) {      // softplus(t) + m = ln2 log2(1 + 2^(t log2 e)) + m, t in [-3, 40]
    const T l0 = v2f{__builtin_amdgcn_logf(__builtin_amdgcn_exp2f(ta.x) + 1.0f), __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(tb.x) + 1.0f)};
    const T l1 = v2f{__builtin_amdgcn_logf(__builtin_amdgcn_exp2f(ta.y) + 1.0f), __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(tb.y) + 1.0f)};
    d0 = vfma(l0, splat<T>(LN2), splat<T>(sc.min_slope));
    d1 = vfma(l1, splat<T>(LN2), splat<T>(sc.min_slope));
  } else {
    // natural units; the product is ROUNDED before knot_slope forms v + |v| (see cond_spline_masked)
    T n0 = v2f{ta.x, tb.x} * LN2, n1 = v2f{ta.y, tb.y} * LN2;
    asm volatile("" : "+v"(n0), "+v"(n1));
    d0 = knot_slope<FAST, T, true>(n0, sc);
    d1 = knot_slope<FAST, T, true>(n1, sc);
  }
  const T ibw = M::rcp(bw);
  const T s = bh * ibw;
  const T st = d1 + d0 - s * 2.0f;
  rqs_bin_eval<INV, FAST, T, false, ARG>(v, x0, y0, bw, bh, ibw, s, st, d0, d1, s, out, ld);
  if (maybe_outside(v, sc.lo, sc.hi)) {
    const auto below = vle(v, sc.lo);          // bin 0 was selected: d0 = slope[0]
    const auto above = vge(v, sc.hi);          // bin K-1 was selected: d1 = slope[K]
    const T lo_out = INV ? M::div(v - sc.lo, d0) + sc.lo : vfma(v - sc.lo, d0, splat<T>(sc.lo));
    const T hi_out = INV ? M::div(v - sc.hi, d1) + sc.hi : vfma(v - sc.hi, d1, splat<T>(sc.hi));
    out = vsel(below, lo_out, out);
    out = vsel(above, hi_out, out);
    if constexpr (ARG) {
      ld = vsel(below, d0, ld);
      ld = vsel(above, d1, ld);
    } else {
      const T ld0 = M::log(d0), ld1 = M::log(d1);
      ld = vsel(below, INV ? -ld0 : ld0, ld);
      ld = vsel(above, INV ? -ld1 : ld1, ld);
    }
  }
}

// ---------------------------------------------------------------------------
// Spline whose 3K+1 parameters were just produced by the conditioner, one set
// per sample, in registers.  The bin is selected with a compare/select chain
// over the running knot (registers cannot be indexed per lane); only the two
// slopes of the selected bin are normalised (2 softplus instead of K+1).
// ---------------------------------------------------------------------------
template <int K, bool INV, bool FAST, class T>
__device__ __forceinline__ void cond_spline(const T (&th)[3 * K + 1], T v,
                                            const SplineConstsT<typename Lanes<T>::real>& sc, T& out, T& ld) {
  using M = Math<FAST>;
  T mw = th[0], mh = th[K];
#pragma unroll
  for (int k = 1; k < K; ++k) { mw = vmax(mw, th[k]); mh = vmax(mh, th[K + k]); }
  T ew[K], eh[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    ew[k] = M::exp(th[k] - mw);
    eh[k] = M::exp(th[K + k] - mh);
  }
  T sw = ew[0], sh = eh[0];
#pragma unroll
  for (int k = 1; k < K; ++k) { sw += ew[k]; sh += eh[k]; }
  const T aw = M::rcp(sw) * sc.span_eff, ah = M::rcp(sh) * sc.span_eff;
  T px = splat<T>(sc.lo), py = splat<T>(sc.lo);          // running knot k
  T wk = vfma(ew[0], aw, splat<T>(sc.min_bin)), hk = vfma(eh[0], ah, splat<T>(sc.min_bin));
  T x0 = px, y0 = py, bw = wk, bh = hk, t0 = th[2 * K], t1 = th[2 * K + 1];
#pragma unroll
  for (int k = 1; k < K; ++k) {
    px += wk;
    py += hk;
    if (k == K - 1) { wk = sc.hi - px; hk = sc.hi - py; }   // last knot is exactly hi
    else { wk = vfma(ew[k], aw, splat<T>(sc.min_bin)); hk = vfma(eh[k], ah, splat<T>(sc.min_bin)); }
    const auto ge = INV ? vge(v, py) : vge(v, px);
    x0 = vsel(ge, px, x0); y0 = vsel(ge, py, y0);
    bw = vsel(ge, wk, bw); bh = vsel(ge, hk, bh);
    t0 = vsel(ge, th[2 * K + k], t0); t1 = vsel(ge, th[2 * K + k + 1], t1);
  }
  const T d0 = knot_slope<FAST, T>(t0, sc), d1 = knot_slope<FAST, T>(t1, sc);
  const T ibw = M::rcp(bw);
  const T s = bh * ibw;
  const T st = d1 + d0 - s * 2.0f;
  rqs_bin_eval<INV, FAST, T, false>(v, x0, y0, bw, bh, ibw, s, st, d0, d1, s, out, ld);
  if (maybe_outside(v, sc.lo, sc.hi)) {
    const auto below = vle(v, sc.lo);          // bin 0 was selected: d0 = slope[0]
    const auto above = vge(v, sc.hi);          // bin K-1 was selected: d1 = slope[K]
    const T lo_out = INV ? M::div(v - sc.lo, d0) + sc.lo : vfma(v - sc.lo, d0, splat<T>(sc.lo));
    const T hi_out = INV ? M::div(v - sc.hi, d1) + sc.hi : vfma(v - sc.hi, d1, splat<T>(sc.hi));
    const T ld0 = M::log(d0), ld1 = M::log(d1);
    out = vsel(below, lo_out, out);
    ld = vsel(below, INV ? -ld0 : ld0, ld);
    out = vsel(above, hi_out, out);
    ld = vsel(above, INV ? -ld1 : ld1, ld);
  }
}

// ---------------------------------------------------------------------------
// Precise position path of the data -> base direction (spline FORWARD; log_prob,
// inverse).  log_prob = sum -x^2/2 + ildj multiplies the error of the recovered
// base point by |x| <= 5, and in plain fp32 that error is ~2e-6: the K softmax
// terms carry ~1e-7 relative error each (v_exp_f32, the sum, the reciprocal) and
// the knots are 20 x their prefix sums.  scripts/numerics/exp_logprob_precision.py
// (float64 / float32 per stage on the CPU): normalisation + knot positions in
// float64 with everything else (conditioner, slopes, log-det) in fp32 gives
// 1e-6 .. 2.5e-6 max |d log_prob| on 65 536 samples; all-fp32 gives 1.1e-5 ..
// 1.5e-5.  So here, per conditioner spline:
//   * e_k = 2^(t_k) to ~1e-9 relative: t = -(idx / 32) + r, 2^(-idx/32) from a
//     1 025-entry float64 table in LDS (t >= -32; smaller terms are below 2^-32
//     of the largest), 2^r - 1 by a cubic in fp32 (|r| <= 1/64);
//   * prefix sums, the total, the quotient S_k / E and the corner
//     x0 = lo + k min_bin + span S_k / E in float64 (v_add_f64 / v_fma_f64 issue
//     at the rate of unpacked fp32 on gfx950; a float-pair emulation costs 6-8
//     fp32 operations per addition);
//   * the offset in the bin (double)v - x0 and the result y0 + increment in
//     float64, rounded once; the bin is chosen on the fp32-rounded knots (next to
//     a knot either neighbour gives the same value: the spline is C1).
// One sample at a time (sample pairs are split: no packed float64 exists).
// ---------------------------------------------------------------------------
constexpr int EXP2_STEPS = 32;                    // table points per unit exponent
constexpr int EXP2_RANGE = 32;                    // exponents in [-32, 0]
constexpr int EXP2_N = EXP2_STEPS * EXP2_RANGE + 1;      // doubles in the table: 2^(-i/32), i = 0 .. 1024
constexpr float LOG2E_LO = 1.92596299112661746e-08f;     // log2(e) - (float)log2(e)

__device__ __forceinline__ double exp2_precise(float t, float t_err, const double* e2tab) {
  t = fmaxf(t, -(float)EXP2_RANGE);
  const unsigned idx = (unsigned)fmaf(t, -(float)EXP2_STEPS, 0.5f);      // v_cvt_u32_f32 truncates: round to nearest
  const float r = fmaf((float)idx, 1.0f / EXP2_STEPS, t) + t_err;       // exact up to t_err: |r| <= 1/64
  const float p = r * fmaf(r, fmaf(r, 0.0555041086648215799f, 0.240226506959100712f), 0.693147180559945309f);
  const double T = e2tab[idx];
  return fma(T, (double)p, T);
}

typedef SplineConstsT<double> PreciseConsts;     // the float64 copy of the spline constants (ModelArgs::scd)

// sum_d x_d^2 of the recovered base point in float64, from fp32 values and what their rounding dropped
template <class T> struct BaseAcc;
template <> struct BaseAcc<float> {
  double b = 0.0;
  __device__ __forceinline__ void add(float o, float lo) { const double x = (double)o + (double)lo; b = fma(x, x, b); }
  __device__ __forceinline__ float log_prob(float ildj, int D) const {
    return (float)(fma(-0.5, b, -(double)D * 0.91893853320467274178) + (double)ildj);
  }
};
template <> struct BaseAcc<v2f> {
  double bx = 0.0, by = 0.0;
  __device__ __forceinline__ void add(v2f o, v2f lo) {
    const double x = (double)o.x + (double)lo.x, y = (double)o.y + (double)lo.y;
    bx = fma(x, x, bx); by = fma(y, y, by);
  }
  __device__ __forceinline__ v2f log_prob(v2f ildj, int D) const {
    const double c = -(double)D * 0.91893853320467274178;
    return v2f{(float)(fma(-0.5, bx, c) + (double)ildj.x), (float)(fma(-0.5, by, c) + (double)ildj.y)};
  }
};
template <> struct BaseAcc<double> {       // (the float64 kernels never take the precise path)
  __device__ __forceinline__ void add(double, double) {}
  __device__ __forceinline__ double log_prob(double ildj, int) const { return ildj; }
};

// 1 / x in float64: hardware fp32 reciprocal + one Newton step (v_rcp_f64 issues at a quarter of the rate)
template <bool FAST> __device__ __forceinline__ double rcp_f64(double x) {
  const double r = (double)Math<FAST>::rcp((float)x);
  return r * fma(-x, r, 2.0);
}

// The forward rational-quadratic map inside one bin in float64, from the bin's corner (x0, y0), width, height and
// the two fp32 slopes; `v + v_lo` is the input as a float pair.  Returns the result as a float pair and log f'.
template <bool FAST>
__device__ __forceinline__ void rqs_fwd_bin_f64(float v, float v_lo, double x0, double y0, double bw, double bh,
                                                float d0, float d1, float& out, float& out_lo, float& ld) {
  using M = Math<FAST>;
  const double ibw = rcp_f64<FAST>(bw);
  const double s = bh * ibw;
  const double z = fmin(fmax((((double)v + (double)v_lo) - x0) * ibw, 0.0), 1.0);
  const double zz = z * z, z1mz = z - zz;
  const double st = ((double)d1 + (double)d0) - 2.0 * s;
  const double den = fma(st, z1mz, s);
  const double iden = rcp_f64<FAST>(den);
  const double out_d = fma(bh * fma(s, zz, (double)d0 * z1mz), iden, y0);
  out = (float)out_d;
  out_lo = (float)(out_d - (double)out);
  // log f' = 2 log s + log(d1 z^2 + 2 s z (1 - z) + d0 (1 - z)^2) - 2 log den: relative accuracy is enough
  const float zf = (float)z, sf = (float)s, omz = 1.0f - zf, z1f = (float)z1mz;
  const float num2 = fmaf(d1, zf * zf, fmaf(2.0f * sf, z1f, d0 * omz * omz));
  const float q = sf * (float)iden;
  ld = M::log(num2 * q * q);
}

// LOG2_UNITS: the 2K softmax logits are already in log2 units and the slope logits carry the softplus offset
// (rows of the conditioner tables, cnf_pwl.h).  The input is the float pair v + v_lo (what the previous layer's
// rounding dropped); returns out (fp32) and what ITS rounding dropped (`out_lo`).
template <int K, bool FAST, bool LOG2_UNITS>
__device__ __forceinline__ void cond_spline_precise(const float (&th)[3 * K + 1], float v, float v_lo,
                                                    const SplineConsts& sc, const PreciseConsts& pc,
                                                    const double* e2tab, float& out, float& ld, float& out_lo) {
  using M = Math<FAST>;
  float mw = th[0], mh = th[K];
#pragma unroll
  for (int k = 1; k < K; ++k) { mw = fmaxf(mw, th[k]); mh = fmaxf(mh, th[K + k]); }
  double Sw[K + 1], Sh[K + 1], ew[K], eh[K];
  Sw[0] = 0.0; Sh[0] = 0.0;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float dw = th[k] - mw, dh = th[K + k] - mh;
    float tw = dw, twe = 0.0f, tg = dh, tge = 0.0f;
    if (!LOG2_UNITS) {
      tw = dw * LOG2E; twe = fmaf(dw, LOG2E, -tw) + dw * LOG2E_LO;
      tg = dh * LOG2E; tge = fmaf(dh, LOG2E, -tg) + dh * LOG2E_LO;
    }
    ew[k] = exp2_precise(tw, twe, e2tab); eh[k] = exp2_precise(tg, tge, e2tab);
    Sw[k + 1] = Sw[k] + ew[k]; Sh[k + 1] = Sh[k] + eh[k];
  }
  const double aw = pc.span_eff * rcp_f64<FAST>(Sw[K]), ah = pc.span_eff * rcp_f64<FAST>(Sh[K]);
  // bin on the fp32-rounded x knots; the selected bin's prefix sums, terms and slope logits by select chains
  const float awf = (float)aw;
  double Sw_sel = 0.0, Sh_sel = 0.0, ew_sel = ew[0], eh_sel = eh[0];
  float t0 = th[2 * K], t1 = th[2 * K + 1];
  int kk = 0;
#pragma unroll
  for (int j = 1; j < K; ++j) {
    const float knot = fmaf((float)Sw[j], awf, sc.lo + (float)j * sc.min_bin);
    const bool ge = v >= knot;
    Sw_sel = ge ? Sw[j] : Sw_sel; Sh_sel = ge ? Sh[j] : Sh_sel;
    ew_sel = ge ? ew[j] : ew_sel; eh_sel = ge ? eh[j] : eh_sel;
    t0 = ge ? th[2 * K + j] : t0; t1 = ge ? th[2 * K + j + 1] : t1;
    kk += ge ? 1 : 0;
  }
  const double base_k = fma((double)kk, pc.min_bin, pc.lo);
  const double x0 = fma(aw, Sw_sel, base_k), y0 = fma(ah, Sh_sel, base_k);
  const double bw = fma(aw, ew_sel, pc.min_bin), bh = fma(ah, eh_sel, pc.min_bin);
  if (LOG2_UNITS) {          // table rows carry (t + offset) log2 e; rounded product (see cond_spline_masked)
    t0 *= LN2; t1 *= LN2;
    asm volatile("" : "+v"(t0), "+v"(t1));
  }
  const float d0 = knot_slope<FAST, float, LOG2_UNITS>(t0, sc), d1 = knot_slope<FAST, float, LOG2_UNITS>(t1, sc);
  rqs_fwd_bin_f64<FAST>(v, v_lo, x0, y0, bw, bh, d0, d1, out, out_lo, ld);
  if (maybe_outside(v, sc.lo, sc.hi)) {      // linear tails: bin 0 / bin K-1 were selected
    if (v <= sc.lo) { out = fmaf(v - sc.lo, d0, sc.lo); ld = M::log(d0); out_lo = 0.0f; }
    if (v >= sc.hi) { out = fmaf(v - sc.hi, d1, sc.hi); ld = M::log(d1); out_lo = 0.0f; }
  }
}

template <int K, bool FAST, bool LOG2_UNITS>
__device__ __forceinline__ void cond_spline_precise(const v2f (&th)[3 * K + 1], v2f v, v2f v_lo, const SplineConsts& sc,
                                                    const PreciseConsts& pc, const double* e2tab,
                                                    v2f& out, v2f& ld, v2f& out_lo) {
  float tx[3 * K + 1], ty[3 * K + 1];
#pragma unroll
  for (int j = 0; j < 3 * K + 1; ++j) { tx[j] = th[j].x; ty[j] = th[j].y; }
  float ox, lx, rx, oy, ly, ry;
  cond_spline_precise<K, FAST, LOG2_UNITS>(tx, v.x, v_lo.x, sc, pc, e2tab, ox, lx, rx);
  cond_spline_precise<K, FAST, LOG2_UNITS>(ty, v.y, v_lo.y, sc, pc, e2tab, oy, ly, ry);
  out = v2f{ox, oy}; ld = v2f{lx, ly}; out_lo = v2f{rx, ry};
}

// The shared `first` spline on the precise path: bin from the fp32 knots, the bin's constants from the float64
// copy of the prepared table (`tabd`, staged in LDS), the in-bin map in float64.
template <int K, bool FAST>
__device__ __forceinline__ void table_spline_precise(const float* tab, const double* tabd, float v, float v_lo,
                                                     const SplineConsts& sc, float& out, float& ld, float& out_lo) {
  const int k = bin_of<K, float>(tab + tab_off(F_XK, K), v);
  const double* r = tabd + k;
  rqs_fwd_bin_f64<FAST>(v, v_lo, r[tab_off(F_X0, K)], r[tab_off(F_Y0, K)], r[tab_off(F_BW, K)], r[tab_off(F_BH, K)],
                        (float)r[tab_off(F_D0, K)], (float)r[tab_off(F_D1, K)], out, out_lo, ld);
  if (maybe_outside(v, sc.lo, sc.hi)) {
    const float* tl = tab + tab_off(F_TAIL, K);
    if (v <= sc.lo) { out = fmaf(v - sc.lo, tl[T_DLO], sc.lo); ld = tl[T_LOG_DLO]; out_lo = 0.0f; }
    if (v >= sc.hi) { out = fmaf(v - sc.hi, tl[T_DHI], sc.hi); ld = tl[T_LOG_DHI]; out_lo = 0.0f; }
  }
}
template <int K, bool FAST>
__device__ __forceinline__ void table_spline_precise(const float* tab, const double* tabd, v2f v, v2f v_lo,
                                                     const SplineConsts& sc, v2f& out, v2f& ld, v2f& out_lo) {
  float ox, lx, rx, oy, ly, ry;
  table_spline_precise<K, FAST>(tab, tabd, v.x, v_lo.x, sc, ox, lx, rx);
  table_spline_precise<K, FAST>(tab, tabd, v.y, v_lo.y, sc, oy, ly, ry);
  out = v2f{ox, oy}; ld = v2f{lx, ly}; out_lo = v2f{rx, ry};
}

// ---------------------------------------------------------------------------
// Conditioner MLP (flows.py:57-84): [c, v_0..v_{d-1}] -> H (relu) -> ... ->
// H (relu) -> P.  `w` is a wave-uniform pointer into the prepared weights, so
// every weight is a scalar (SGPR) operand of the per-lane (packed) FMA: no LDS
// or VGPR traffic for weights at all.  The d inputs v_q are read from this
// thread's own LDS column.
// ---------------------------------------------------------------------------

// Pins every element in a VGPR at this point of the program: without it the
// compiler sinks whole accumulation chains (and the 16x16 weights they need,
// as spilled SGPRs) down to their first use in the spline code.
template <int N> __device__ __forceinline__ void materialize(float (&v)[N]) {
#pragma unroll
  for (int j = 0; j < N; ++j) asm volatile("" : "+v"(v[j]));
}
template <int N> __device__ __forceinline__ void materialize(v2f (&v)[N]) {
#pragma unroll
  for (int j = 0; j < N; ++j) asm volatile("" : "+v"(v[j]));
}

template <int N> __device__ __forceinline__ void materialize(double (&v)[N]) {
#pragma unroll
  for (int j = 0; j < N; ++j) asm volatile("" : "+v"(v[j]));
}

template <int N> __device__ __forceinline__ void load_row(uniform_ptr p, float (&v)[N]) {
#pragma unroll
  for (int j = 0; j < N; ++j) v[j] = p[j];
}

// acc[j] += sum_i in[i] * W[i][j] for a wave-uniform row-major W[R][N].
// One row = one s_load_dwordx16 into SGPRs + N per-lane (packed) FMAs with an
// SGPR operand.  Two things are controlled by hand with scheduling barriers:
//  * hipcc would cluster ALL of the layer's scalar loads at the top (~600 SGPRs
//    -> spilled to VGPR lanes); here at most two rows are live;
//  * scalar loads return out of order, so every wait is lgkmcnt(0): to overlap
//    the next row's load with this row's FMAs the wait for THIS row must come
//    first.  Order per row: [first FMA: waits for row i] [issue load of row
//    i+1] [remaining FMAs].
template <int R, int N, int G, class T>
__device__ __forceinline__ void dense_acc(uniform_ptr W, const T (&in)[R], T (&acc)[N]) {
  float cur[N];
#pragma unroll
  for (int t = 0; t < N; ++t) cur[t] = W[t];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    acc[0] = vfma(cur[0], in[i], acc[0]);
    __builtin_amdgcn_sched_barrier(0);
    float nxt[N];
    if (i + 1 < R) {
#pragma unroll
      for (int t = 0; t < N; ++t) nxt[t] = W[(i + 1) * N + t];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 1; j < N; ++j) acc[j] = vfma(cur[j], in[i], acc[j]);
    __builtin_amdgcn_sched_barrier(0);
    if (i + 1 < R) {
#pragma unroll
      for (int t = 0; t < N; ++t) cur[t] = nxt[t];
    }
  }
}

constexpr int row_group(int R, int N) { return 1; }

// this thread's value(s) of input dimension `idx` in its LDS column
template <class T> __device__ __forceinline__ T lds_get(const typename Lanes<T>::real* col, int idx, int stride);
template <> __device__ __forceinline__ float lds_get<float>(const float* col, int idx, int stride) {
  return col[idx * stride];
}
template <> __device__ __forceinline__ double lds_get<double>(const double* col, int idx, int stride) {
  return col[idx * stride];
}
__device__ __forceinline__ void lds_put(double* col, int idx, int stride, double v) { col[idx * stride] = v; }
template <> __device__ __forceinline__ v2f lds_get<v2f>(const float* col, int idx, int stride) {
  return *reinterpret_cast<const v2f*>(col + idx * stride);
}
__device__ __forceinline__ void lds_put(float* col, int idx, int stride, float v) { col[idx * stride] = v; }
__device__ __forceinline__ void lds_put(float* col, int idx, int stride, v2f v) {
  *reinterpret_cast<v2f*>(col + idx * stride) = v;
}

// sin / cos of a conditioner input for the periodized model (flows.py:58-64): ocml's full-accuracy functions in
// every math mode -- two per input next to a 2 (1 + d) x H matrix product
__device__ __forceinline__ void vsincos(float x, float& s, float& c) { s = sinf(x); c = cosf(x); }
__device__ __forceinline__ void vsincos(double x, double& s, double& c) { s = ::sin(x); c = ::cos(x); }
__device__ __forceinline__ void vsincos(v2f x, v2f& s, v2f& c) {
  s = v2f{sinf(x.x), sinf(x.y)}; c = v2f{cosf(x.x), cosf(x.y)};
}

// PERIODIC: the first linear layer has 2 (1 + d) rows and sees [sin(c), sin(v_0) .., cos(c), cos(v_0) ..]
// (jnp.concatenate([sin(x)], [cos(x)]) of x = [c, v], flows.py:58-64 with num_fourier_feat = 1).
template <int H, int P, class T, bool PERIODIC = false>
__device__ __forceinline__ void conditioner(uniform_ptr w, int d, int M, T c, const typename Lanes<T>::real* col,
                                            int first_idx, int idx_step, int stride, T (&th)[P]) {
  T h[H];
  w = launder(w);
  uniform_ptr b0 = w + (PERIODIC ? 2 : 1) * (1 + d) * H;
  if constexpr (PERIODIC) {
    T sn, cs;
    vsincos(c, sn, cs);
    {
      float ws[H], wc[H], bb[H];
      load_row<H>(w, ws);
      load_row<H>(w + (1 + d) * H, wc);
      load_row<H>(b0, bb);
#pragma unroll
      for (int j = 0; j < H; ++j) h[j] = vfma(wc[j], cs, vfma(ws[j], sn, splat<T>(bb[j])));
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int q = 0; q < d; ++q) {
      const T v = lds_get<T>(col, first_idx + q * idx_step, stride);
      vsincos(v, sn, cs);
      float ws[H], wc[H];
      load_row<H>(w + (1 + q) * H, ws);
      load_row<H>(w + (2 + d + q) * H, wc);
#pragma unroll
      for (int j = 0; j < H; ++j) h[j] = vfma(wc[j], cs, vfma(ws[j], sn, h[j]));
    }
  } else {
    {
      float wc[H], bb[H];
      load_row<H>(w, wc);
      load_row<H>(b0, bb);
#pragma unroll
      for (int j = 0; j < H; ++j) h[j] = vfma(wc[j], c, splat<T>(bb[j]));
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int q = 0; q < d; ++q) {              // runtime loop: d is not a template arg
      const T v = lds_get<T>(col, first_idx + q * idx_step, stride);
      float wr[H];
      load_row<H>(w + (1 + q) * H, wr);
#pragma unroll
      for (int j = 0; j < H; ++j) h[j] = vfma(wr[j], v, h[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < H; ++j) h[j] = vrelu(h[j]);
  materialize<H>(h);
  w = b0 + H;
  for (int m = 1; m < M; ++m) {
    T g[H];
    uniform_ptr b = w + H * H;
    {
      float bb[H];
      load_row<H>(b, bb);
#pragma unroll
      for (int j = 0; j < H; ++j) g[j] = splat<T>(bb[j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    dense_acc<H, H, row_group(H, H), T>(w, h, g);
#pragma unroll
    for (int j = 0; j < H; ++j) h[j] = vrelu(g[j]);
    materialize<H>(h);
    w = b + H;
  }
  {
    float bb[P];
    load_row<P>(w + H * P, bb);
#pragma unroll
    for (int j = 0; j < P; ++j) th[j] = splat<T>(bb[j]);
  }
  __builtin_amdgcn_sched_barrier(0);
  dense_acc<H, P, row_group(H, P), T>(w, h, th);
  materialize<P>(th);
}

// ---------------------------------------------------------------------------
// MFMA conditioner (H = 16, P = 16 only: the reference's network).  The two
// 16x16 matmuls of the MLP run on the matrix cores as exact-fp32
// v_mfma_f32_16x16x4_f32, freeing ~80 % of the conditioner's VALU slots for the
// spline code of the other waves (VALU and MFMA pipes issue concurrently).
//
// Layouts (lane l: g = l >> 4, s = l & 15):
//   * sample groups: group q is the 16 samples owned by lanes 16*(q / N) + s,
//     component q % N (N = samples per lane): Q = 4*N groups per wave;
//   * D = W^T * H^T per group: samples on the N axis.  MFMA operand maps
//     (checked on hardware by scripts/probes/mfma_probe.hip): A lane (g,i) holds
//     A[i][k=g], B lane (g,s) holds B[k=g][n=s], result lane (g,s) register r
//     holds D[row=4g+r][col=s];
//   * the k index of step t is permuted to k = 4g + t, so that a layer's
//     result registers ARE the next layer's B operands (register r = step t):
//     chained layers need no lane movement at all.  A-operands and biases are
//     pre-permuted accordingly by prepare_kernel (`wq`, one float4 per lane
//     per field: W0 rows, b0, then {A, bias} per dense layer);
//   * entry: the first layer (K = 1+d inputs) is evaluated directly in the B
//     layout; a lane fetches c and v of the group's sample with ds_bpermute;
//   * exit: the 16 spline parameters of a sample sit in 4 lanes x 4 registers;
//     a 4x4 transpose over lane blocks (2 x v_permlane32_swap +
//     2 x v_permlane16_swap per register quad) brings them to the sample's lane.
// ---------------------------------------------------------------------------
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float comp_of(float v, int) { return v; }
__device__ __forceinline__ float comp_of(v2f v, int i) { return i == 0 ? v.x : v.y; }

__device__ __forceinline__ float lane_fetch(int byte_addr, float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v)));
}

// in-place 4x4 transpose between register index b and lane block g
__device__ __forceinline__ void transpose4(float& x0, float& x1, float& x2, float& x3) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x0), __float_as_uint(x2), false, false);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x1), __float_as_uint(x3), false, false);
  auto c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
  auto d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
  x0 = __uint_as_float(c[0]); x1 = __uint_as_float(c[1]);
  x2 = __uint_as_float(d[0]); x3 = __uint_as_float(d[1]);
}

__device__ __forceinline__ void set_comp(float& dst, int, float v) { dst = v; }
__device__ __forceinline__ void set_comp(v2f& dst, int i, float v) { if (i == 0) dst.x = v; else dst.y = v; }

// lane-per-sample layout (16 values per lane) <-> MFMA operand layout (lane (g, s), register r: unit 4g + r of
// sample 16q + s of group q): the same 4x4 permlane transpose both ways
__device__ __forceinline__ void to_mfma_layout(const float (&v)[16], float (&m)[4][4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x0 = v[r], x1 = v[4 + r], x2 = v[8 + r], x3 = v[12 + r];
    transpose4(x0, x1, x2, x3);
    m[0][r] = x0; m[1][r] = x1; m[2][r] = x2; m[3][r] = x3;
  }
}
__device__ __forceinline__ void from_mfma_layout(const float (&m)[4][4], float (&v)[16]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x0 = m[0][r], x1 = m[1][r], x2 = m[2][r], x3 = m[3][r];
    transpose4(x0, x1, x2, x3);
    v[r] = x0; v[4 + r] = x1; v[8 + r] = x2; v[12 + r] = x3;
  }
}

// First hidden layer in the lane-per-sample layout with SCALAR weights (the 16 (1 + d) FMAs are the same count in
// either layout, but the MFMA-layout form reads one weight vector per input row from global memory inside a
// runtime loop: d + 2 dependent ~1 us round trips per conditioner at dim 10, against scalar-cache loads the
// scheduler issues ahead), then one 4x4 permlane transpose into the MFMA operand layout.
template <class T>
__device__ __forceinline__ void first_layer_lane(uniform_ptr w, int d, T c, const float* col, int first_idx,
                                                 int idx_step, int stride, T (&h1)[16]) {
  w = launder(w);
  uniform_ptr b0 = w + (1 + d) * 16;
  {
    float wc[16], bb[16];
    load_row<16>(w, wc); load_row<16>(b0, bb);
#pragma unroll
    for (int j = 0; j < 16; ++j) h1[j] = vfma(wc[j], c, splat<T>(bb[j]));
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int q = 0; q < d; ++q) {
    const T v = lds_get<T>(col, first_idx + q * idx_step, stride);
    float wr[16];
    load_row<16>(w + (1 + q) * 16, wr);
#pragma unroll
    for (int j = 0; j < 16; ++j) h1[j] = vfma(wr[j], v, h1[j]);
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) h1[j] = vrelu(h1[j]);
}

// `wflat` (optional): the conditioner's weights in the flat layout; with it the first hidden layer is evaluated in
// the lane layout with scalar weights and transposed (first_layer_lane above) -- no dependent vector loads per
// input row.
template <class T>
__device__ __forceinline__ void conditioner_mfma(const f4* __restrict__ wq, int d, int M, T c,
                                                 const float* col, int first_idx, int idx_step,
                                                 int stride, T (&th)[16], uniform_ptr wflat = nullptr) {
  constexpr int N = Lanes<T>::N;
  constexpr int Q = 4 * N;
  const int lane = threadIdx.x & 63;
  const int s15 = lane & 15;
  float h[Q][4];
  if (wflat) {
    T h1[16];
    first_layer_lane<T>(wflat, d, c, col, first_idx, idx_step, stride, h1);
#pragma unroll
    for (int n = 0; n < N; ++n) {
      float hv[16], m[4][4];
#pragma unroll
      for (int j = 0; j < 16; ++j) hv[j] = comp_of(h1[j], n);
      to_mfma_layout(hv, m);
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
#pragma unroll
        for (int r = 0; r < 4; ++r) h[qq * N + n][r] = m[qq][r];
      }
    }
  } else {
    const f4 w0c = wq[lane];
    const f4 b0 = wq[(1 + d) * 64 + lane];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const float cq = lane_fetch(4 * (16 * (q / N) + s15), comp_of(c, q % N));
#pragma unroll
      for (int t = 0; t < 4; ++t) h[q][t] = fmaf(w0c[t], cq, b0[t]);
    }
  }
  for (int row = 0; row < d && !wflat; ++row) {          // runtime loop over the d conditioning inputs
    const T v = lds_get<T>(col, first_idx + row * idx_step, stride);
    const f4 w = wq[(1 + row) * 64 + lane];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const float vq = lane_fetch(4 * (16 * (q / N) + s15), comp_of(v, q % N));
#pragma unroll
      for (int t = 0; t < 4; ++t) h[q][t] = fmaf(w[t], vq, h[q][t]);
    }
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) {
#pragma unroll
    for (int t = 0; t < 4; ++t) h[q][t] = fmaxf(h[q][t], 0.0f);
  }
  const f4* p = wq + (2 + d) * 64;
  f4 acc[Q];
  f4 A_next = p[lane], bias_next = p[64 + lane];      // the next layer's operands are fetched one layer ahead:
  for (int m = 1; m <= M; ++m) {               // M-1 hidden 16x16 layers + the 16x16 output layer
    const f4 A = A_next, bias = bias_next;     // a ~1 us round trip otherwise waited for at every layer
    p += 128;
    if (m < M) { A_next = p[lane]; bias_next = p[64 + lane]; }
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = bias;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int q = 0; q < Q; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[t], h[q][t], acc[q], 0, 0, 0);
    }
    if (m < M) {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
#pragma unroll
        for (int t = 0; t < 4; ++t) h[q][t] = fmaxf(acc[q][t], 0.0f);
      }
    }
  }
  // exit: acc[q][r] = theta[4g + r] of group q's sample s  ->  th[j] of the lane's own sample(s)
#pragma unroll
  for (int n = 0; n < N; ++n) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x0 = acc[0 * N + n][r], x1 = acc[1 * N + n][r], x2 = acc[2 * N + n][r], x3 = acc[3 * N + n][r];
      transpose4(x0, x1, x2, x3);              // x_g = theta[4g + r] of this lane's sample (component n)
      set_comp(th[0 + r], n, x0); set_comp(th[4 + r], n, x1);
      set_comp(th[8 + r], n, x2); set_comp(th[12 + r], n, x3);
    }
  }
}

// floats of one conditioner's MFMA-layout block: (1+d) W0 rows, b0, M x {A, bias}, 64 lanes x 4
__host__ __device__ constexpr int64_t cond_floats_mfma(int d, int M) { return 256 * (int64_t)((1 + d) + 1 + 2 * M); }

// floats of one conditioner whose first linear layer has `rows` inputs
__host__ __device__ constexpr int64_t cond_floats_rows(int rows, int H, int M, int P) {
  return (int64_t)rows * H + H + (int64_t)(M - 1) * (H * H + H) + (int64_t)H * P + P;
}
__host__ __device__ constexpr int64_t cond_floats(int d, int H, int M, int P) { return cond_floats_rows(1 + d, H, M, P); }
// periodized (flows.py:58-64): the MLP sees [sin(c, v), cos(c, v)]: 2 (1 + d) rows
__host__ __device__ inline int64_t cond_floats_p(int d, int H, int M, int P, bool periodic) {
  return cond_floats_rows((periodic ? 2 : 1) * (1 + d), H, M, P);
}

// A wave's contribution to the per-slice sums of a loss kernel: the tiles' wave-reduced partial sums are added up
// in a register while the wave stays in one slice, and go to sums[slice] with ONE double atomic when the slice
// changes and at the end.  (One atomic per wave and tile put 65 536 of them on the single address of a 4.2 M-sample
// slice -- a compare-and-swap loop each, without -munsafe-fp-atomics -- and cost more than the flow passes.)
struct SliceSum {
  double run = 0.0;
  int64_t slice = -1;
  // (both are wave-uniform: kept in scalar registers -- four vector registers the backward kernels have no room for)
  static __device__ __forceinline__ double uniform(double x) {
    const long long b = __double_as_longlong(x);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
  }
  double* dst = nullptr;      // the sums array `slice` belongs to (a launch may serve several terms)
  __device__ __forceinline__ void flush() {
    if (slice >= 0 && (threadIdx.x & 63) == 0 && run != 0.0) unsafeAtomicAdd(dst + slice, run);
    run = 0.0;
  }
  __device__ __forceinline__ void flush(double* sums) { (void)sums; flush(); }
  // s: the tile's slice (wave-uniform); part: the tile's sum over the wave's lanes (the same in every lane)
  __device__ __forceinline__ void add(double* sums, int64_t s, float part) {
    if (s != slice || sums != dst) { flush(); slice = s; dst = sums; }
    run = uniform(run + (double)part);
  }
};

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

// Counter block `blk` of the base-noise stream of `seed` -> its four standard
// normals (Philox4x32-10, two Box-Muller pairs from 24-bit uniforms).  Element
// e of the stream is z[e & 3] of block e >> 2: cnf_fill_normal, the in-kernel
// draws of the loss kernels and the oracle all go through this definition.
__device__ __forceinline__ void philox_normals4(uint64_t seed, uint64_t blk, float (&z)[4]) {
  uint32_t u[4];
  philox4x32((uint32_t)blk, (uint32_t)(blk >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), u);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float u1 = (float)((u[2 * p] >> 8) + 1u) * (1.0f / 16777216.0f);
    const float u2 = (float)(u[2 * p + 1] >> 8) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincospif(2.0f * u2, &sn, &cs);
    z[2 * p] = rad * cs;
    z[2 * p + 1] = rad * sn;
  }
}

}  // namespace cnf
