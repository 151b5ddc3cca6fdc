// cnf_backward.h -- reverse-mode derivatives of the conditional RQS flow for the
// reference's network (hidden 16, 2 hidden layers, 5 bins => P = 16), one
// sample per lane.  Replaces jax.value_and_grad over cnf_ot/mfc/solvers.py:94.
//
// The weight gradients are batch reductions  dW[i][j] = sum_samples h_i * g_j :
// GEMMs whose K dimension is the batch, so they run on the matrix cores
// (v_mfma_f32_16x16x4_f32, exact fp32): the per-sample operands are staged
// through LDS into MFMA layout, 4 samples per instruction.
#pragma once

#include "cnf_common.h"

namespace cnf {

// ---------------------------------------------------------------------------
// Partials of the forward spline map f(x; bin) and of its log-derivative
// ld(x; bin) = log f'(x) with respect to x and the bin's six quantities
// (x0, y0, bw, bh, d0, d1).  f_y0 = 1 and ld_y0 = 0 are implicit.
// Notation of SURVEY.md Appendix A: z = (x-x0)/bw, s = bh/bw, q = z(1-z),
// den = s + (d0+d1-2s) q, N = s z^2 + d0 q, f = y0 + bh N / den,
// A = d1 z^2 + 2 s q + d0 (1-z)^2, ld = 2 log s + log A - 2 log den.
// ---------------------------------------------------------------------------
struct BinPartials {
  float f_x, f_x0, f_bw, f_bh, f_d0, f_d1;
  float l_x, l_x0, l_bw, l_bh, l_d0, l_d1;
  float f_y0;      // 1 inside the range; 0 on the linear tails, which hang on the fixed corners (lo, lo) / (hi, hi)
};

// FAST: v_rcp_f32 (1 ulp) for the six reciprocals instead of IEEE division (ten instructions each)
// rqs_partials_at: the position in the bin given as the pair (z, omz = 1 - z) -- a caller that knows the small one of
// the two to full relative precision (cond_spline_bwd's inverse) passes it as such; x only decides the tails.
template <bool FAST = false>
__device__ __forceinline__ BinPartials rqs_partials_at(float z, float omz, float x, float bw, float bh, float d0,
                                                       float d1, float lo, float hi) {
  using M = Math<FAST>;
  BinPartials p;
  const float ibw = M::rcp(bw);
  const float s = bh * ibw;
  const float q = z * omz, qp = omz - z;
  const float st = d0 + d1 - 2.0f * s;
  const float Nn = fmaf(s * z, z, d0 * q);
  const float den = fmaf(st, q, s);
  const float iden = M::rcp(den);
  const float A = fmaf(d1 * z, z, fmaf(2.0f * s, q, d0 * omz * omz));
  const float iA = M::rcp(A);
  const float iden2 = iden * iden;
  p.f_x = s * s * A * iden2;
  // (forms without cancellation as q -> 0:  z^2 den - N (1 - 2q) = q (z^2 st - d0 + 2N),  den - N = omz (s omz + d1 z),
  //  2/s - 2 (1 - 2q)/den = 2q (d0 + d1)/(s den) -- the differences lost two digits within 1e-4 of a knot)
  const float f_s = bh * q * (fmaf(z * z, st, -d0) + 2.0f * Nn) * iden2;
  p.f_d0 = bh * q * omz * fmaf(s, omz, d1 * z) * iden2;
  p.f_d1 = -bh * Nn * q * iden2;
  p.f_x0 = -p.f_x;
  p.f_bw = -p.f_x * z - f_s * s * ibw;
  p.f_bh = Nn * iden + f_s * ibw;
  const float Az = 2.0f * (d1 * z + s * qp - d0 * omz);
  const float denz = st * qp;
  const float l_z = Az * iA - 2.0f * denz * iden;
  const float l_s = 2.0f * q * fmaf((d0 + d1) * iden, M::rcp(s), iA);
  p.l_x = l_z * ibw;
  p.l_x0 = -p.l_x;
  p.l_bw = -(l_z * z + l_s * s) * ibw;
  p.l_bh = l_s * ibw;
  p.l_d0 = omz * omz * iA - 2.0f * q * iden;
  p.l_d1 = z * z * iA - 2.0f * q * iden;
  p.f_y0 = 1.0f;
  if (x <= lo) {        // linear tail through (lo, lo) with slope d0 of bin 0
    p = BinPartials{d0, 0.f, 0.f, 0.f, x - lo, 0.f, 0.f, 0.f, 0.f, 0.f, M::rcp(d0), 0.f, 0.f};
  }
  if (x >= hi) {        // linear tail through (hi, hi) with slope d1 of the last bin
    p = BinPartials{d1, 0.f, 0.f, 0.f, 0.f, x - hi, 0.f, 0.f, 0.f, 0.f, 0.f, M::rcp(d1), 0.f};
  }
  return p;
}

// Adjoints of the bin quantities from the adjoints (o_bar, l_bar) of the
// spline's (output, logdet).  INV=false: out = f(v).  INV=true: out = f^-1(v),
// logdet = -ld(out): by the implicit function theorem, with the partials taken
// at x = out,  v_bar = (o_bar - l_bar l_x) / f_x,  p_bar = -(f_p v_bar + l_bar l_p).
struct BinAdjoint { float v, x0, y0, bw, bh, d0, d1; };

template <bool INV, bool FAST = false>
__device__ __forceinline__ BinAdjoint bin_adjoint(const BinPartials& p, float o_bar, float l_bar) {
  BinAdjoint a;
  if (!INV) {
    a.v = o_bar * p.f_x + l_bar * p.l_x;
    a.x0 = o_bar * p.f_x0 + l_bar * p.l_x0;
    a.y0 = o_bar * p.f_y0;      // (the upper tail does not move with the last knot's y: found by scripts/soak_xbar_oracle.py)
    a.bw = o_bar * p.f_bw + l_bar * p.l_bw;
    a.bh = o_bar * p.f_bh + l_bar * p.l_bh;
    a.d0 = o_bar * p.f_d0 + l_bar * p.l_d0;
    a.d1 = o_bar * p.f_d1 + l_bar * p.l_d1;
  } else {
    a.v = (o_bar - l_bar * p.l_x) * Math<FAST>::rcp(p.f_x);
    a.x0 = -(p.f_x0 * a.v + l_bar * p.l_x0);
    a.y0 = -a.v * p.f_y0;
    a.bw = -(p.f_bw * a.v + l_bar * p.l_bw);
    a.bh = -(p.f_bh * a.v + l_bar * p.l_bh);
    a.d0 = -(p.f_d0 * a.v + l_bar * p.l_d0);
    a.d1 = -(p.f_d1 * a.v + l_bar * p.l_d1);
  }
  return a;
}

template <bool FAST = false>
__device__ __forceinline__ BinPartials rqs_partials(float x, float x0, float bw, float bh, float d0,
                                                    float d1, float lo, float hi) {
  const float z = clip01((x - x0) * Math<FAST>::rcp(bw));
  return rqs_partials_at<FAST>(z, 1.0f - z, x, bw, bh, d0, d1, lo, hi);
}

// Partials of the INVERSE map at the point it maps v to, formed here from the bin's own quantities (not taken from a
// forward pass, whose output can lie a rounding outside this bin next to a knot).  The quadratic is solved from the
// nearer end of the bin (the spline is symmetric under z <-> 1 - z, d0 <-> d1, dy <-> bh - dy), so the SMALL one of
// z and 1 - z comes out to full relative precision: where the inverse saturates against a knot, 1 - z recovered from a
// float32 output is good to a few per cent only, and the log-det partials divide by it (scripts/debug_xbar_tail.py).
template <bool FAST = false>
__device__ __forceinline__ BinPartials rqs_partials_inv(float v, float x0, float y0, float bw, float bh, float d0,
                                                        float d1, float lo, float hi) {
  using M = Math<FAST>;
  const float sl = bh * M::rcp(bw), st = d0 + d1 - 2.0f * sl;
  const float dy = fminf(fmaxf(v - y0, 0.0f), bh), dyt = bh - dy;
  const bool low = dy <= dyt;
  const float t = low ? dy : dyt, da = low ? d0 : d1;
  const float c = -sl * t, b = fmaf(-st, t, da * bh), a2 = fmaf(sl, bh, -b);
  const float disc = fmaf(b, b, a2 * c * -4.0f);
  const float r = clip01(M::div(c * -2.0f, b + M::sqrt(disc)));
  const float z = low ? r : 1.0f - r, omz = low ? 1.0f - r : r;
  float out = low ? fmaf(bw, r, x0) : fmaf(-bw, r, x0 + bw);
  if (v <= lo) out = fmaf(v - lo, M::rcp(d0), lo);      // rqs_bin_eval's linear tails
  if (v >= hi) out = fmaf(v - hi, M::rcp(d1), hi);
  return rqs_partials_at<FAST>(z, omz, out, bw, bh, d0, d1, lo, hi);
}

// ---------------------------------------------------------------------------
// Backward of the spline of the shared `first` parameters.  The table holds
// the normalised knots; the softmax / softplus Jacobians are linear in the
// per-bin adjoint sums, so a lane only accumulates  Wb[j] (adjoint of width j),
// Hb[j], Db[j] (adjoint of slope j);  grad_finish_kernel applies the Jacobians
// once, in float64.
// ---------------------------------------------------------------------------
template <int K, bool INV, bool FAST = false>
__device__ __forceinline__ float table_spline_bwd(const float* tab, float v, float out, float o_bar, float l_bar,
                                                  const SplineConsts& sc, float (&Wb)[K], float (&Hb)[K],
                                                  float (&Db)[K + 1]) {
  const float* pos = tab + tab_off(INV ? F_YK : F_XK, K);
  const int k = bin_of<K>(pos, v);
  (void)out;      // (INV: the output is formed here, on this bin's knots, like cond_spline_bwd's: ADVICE r2)
  const float x0 = gather<K>(tab, F_X0, k), bw = gather<K>(tab, F_BW, k), bh = gather<K>(tab, F_BH, k);
  const float d0 = gather<K>(tab, F_D0, k), d1 = gather<K>(tab, F_D1, k);
  BinPartials p;
  if constexpr (INV) p = rqs_partials_inv<FAST>(v, x0, gather<K>(tab, F_Y0, k), bw, bh, d0, d1, sc.lo, sc.hi);
  else p = rqs_partials<FAST>(v, x0, bw, bh, d0, d1, sc.lo, sc.hi);
  const BinAdjoint a = bin_adjoint<INV, FAST>(p, o_bar, l_bar);
  // The bin's adjoints into the per-lane accumulators by 0 / 1 masks in FMAs (was: two compare + select pairs per
  // accumulator, 100 vector instructions per spline; now ~45): ge[j] = [k >= j] is one clamped subtraction each.
  const float kf = (float)k;
  float ge[K + 2];
  ge[0] = 1.0f; ge[K + 1] = 0.0f;
#pragma unroll
  for (int j = 1; j <= K; ++j) ge[j] = clip01(kf - (float)(j - 1));
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const float eq = ge[j] - ge[j + 1];                      // [k == j]
    Wb[j] = fmaf(ge[j + 1], a.x0, fmaf(eq, a.bw, Wb[j]));    // knots left of the bin move with x0, the bin's own width with bw
    Hb[j] = fmaf(ge[j + 1], a.y0, fmaf(eq, a.bh, Hb[j]));
  }
#pragma unroll
  for (int j = 0; j <= K; ++j) {
    const float eq = ge[j] - ge[j + 1], eq1 = j > 0 ? ge[j - 1] - ge[j] : 0.0f;      // [k == j], [k + 1 == j]
    Db[j] = fmaf(eq, a.d0, fmaf(eq1, a.d1, Db[j]));
  }
  return a.v;
}

// ---------------------------------------------------------------------------
// Backward of a conditioner-parameterised spline: recomputes the
// normalisation of cond_spline (same selection), returns the adjoint of the
// spline input and writes theta_bar[3K+1].
// ---------------------------------------------------------------------------
// SLOPES_OUT: the two non-zero slope adjoints and their position instead of the 2K .. 3K entries of tb (which are
// then left untouched): theta_bar[2K + kk] = sb0, theta_bar[2K + kk + 1] = sb1.
// form_out (INV only, wave-uniform): `out` is not taken from the caller -- the inverse map is evaluated here, in the bin
// this function has selected, on ITS knots.  The backward kernels all ask for it: the forward pass normalises the
// softmax in another order, so next to a knot its output can lie a rounding outside the bin selected here, the
// position clips to 0 or 1 and the implicit-function quotient 1 / f' is garbage (found by the table-backward soak:
// both kernels 3e9 off the float64 oracle on two samples of 36 000, profiles/r02_experiments/soak_vjp_case_1_16.log).
template <int K, bool INV, bool FAST, bool SLOPES_OUT = false>
__device__ __forceinline__ float cond_spline_bwd(const float (&th)[3 * K + 1], float v, float out, float o_bar,
                                                 float l_bar, const SplineConsts& sc, float (&tb)[3 * K + 1],
                                                 int* kk_out = nullptr, float* sb0 = nullptr, float* sb1 = nullptr,
                                                 bool form_out = false) {
  using M = Math<FAST>;
  float mw = th[0], mh = th[K];
#pragma unroll
  for (int k = 1; k < K; ++k) { mw = fmaxf(mw, th[k]); mh = fmaxf(mh, th[K + k]); }
  float pw[K], ph[K];
  float sw = 0.0f, sh = 0.0f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    pw[k] = M::exp(th[k] - mw); ph[k] = M::exp(th[K + k] - mh);
    sw += pw[k]; sh += ph[k];
  }
  const float isw = M::rcp(sw), ish = M::rcp(sh);
#pragma unroll
  for (int k = 0; k < K; ++k) { pw[k] *= isw; ph[k] *= ish; }      // softmax probabilities
  float px = sc.lo, py = sc.lo;
  float wk = fmaf(pw[0], sc.span_eff, sc.min_bin), hk = fmaf(ph[0], sc.span_eff, sc.min_bin);
  float x0 = px, y0 = py, bw = wk, bh = hk, t0 = th[2 * K], t1 = th[2 * K + 1];
  float cumw = 0.0f, cumh = 0.0f, pkw = pw[0], pkh = ph[0], cw = 0.0f, ch = 0.0f;
  int kk = 0;
#pragma unroll
  for (int k = 1; k < K; ++k) {
    px += wk; py += hk;
    cw += pw[k - 1]; ch += ph[k - 1];
    if (k == K - 1) { wk = sc.hi - px; hk = sc.hi - py; }
    else { wk = fmaf(pw[k], sc.span_eff, sc.min_bin); hk = fmaf(ph[k], sc.span_eff, sc.min_bin); }
    const bool ge = INV ? (v >= py) : (v >= px);
    x0 = ge ? px : x0; y0 = ge ? py : y0; bw = ge ? wk : bw; bh = ge ? hk : bh;
    t0 = ge ? th[2 * K + k] : t0; t1 = ge ? th[2 * K + k + 1] : t1;
    cumw = ge ? cw : cumw; cumh = ge ? ch : cumh; pkw = ge ? pw[k] : pkw; pkh = ge ? ph[k] : pkh;
    kk += ge ? 1 : 0;
  }
  const float d0 = knot_slope<FAST, float>(t0, sc), d1 = knot_slope<FAST, float>(t1, sc);
  BinPartials p;
  if (INV && form_out) {
    p = rqs_partials_inv<FAST>(v, x0, y0, bw, bh, d0, d1, sc.lo, sc.hi);
  } else {
    p = rqs_partials<FAST>(INV ? out : v, x0, bw, bh, d0, d1, sc.lo, sc.hi);
  }
  const BinAdjoint a = bin_adjoint<INV, FAST>(p, o_bar, l_bar);
  // widths: w_j = span p_j + min_bin, x0 = lo + sum_{j<k} w_j, bw = w_k
  //   theta_bar_j = span p_j (wbar_j - sum_i wbar_i p_i)
  const float Sw = a.x0 * cumw + a.bw * pkw, Sh = a.y0 * cumh + a.bh * pkh;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const float wb = (j < kk ? a.x0 : 0.0f) + (j == kk ? a.bw : 0.0f);
    const float hb = (j < kk ? a.y0 : 0.0f) + (j == kk ? a.bh : 0.0f);
    tb[j] = sc.span_eff * pw[j] * (wb - Sw);
    tb[K + j] = sc.span_eff * ph[j] * (hb - Sh);
  }
  // slopes: d = softplus(t + off) + m  =>  dd/dt = sigmoid(t + off)
  const float sg0 = M::rcp(1.0f + M::exp(-(t0 + sc.sp_offset)));
  const float sg1 = M::rcp(1.0f + M::exp(-(t1 + sc.sp_offset)));
  if constexpr (SLOPES_OUT) {
    *kk_out = kk; *sb0 = a.d0 * sg0; *sb1 = a.d1 * sg1;
  } else {
#pragma unroll
    for (int j = 0; j <= K; ++j) tb[2 * K + j] = (j == kk ? a.d0 * sg0 : 0.0f) + (j == kk + 1 ? a.d1 * sg1 : 0.0f);
  }
  return a.v;
}

// ---------------------------------------------------------------------------
// The same derivatives for SAMPLE PAIRS (two samples per lane, v2f): everything but the transcendentals and the
// table reads is one packed instruction for both samples (vjp_pwl_kernel, the table backward of dim 2).  Same
// formulas as above; where the single-sample forms select with compares (the bin, the tails) these use the 0 / 1
// masks of cond_spline_masked in packed FMAs.
// ---------------------------------------------------------------------------
struct BinPartials2 {
  v2f f_x, f_x0, f_bw, f_bh, f_d0, f_d1;
  v2f l_x, l_x0, l_bw, l_bh, l_d0, l_d1;
  v2f f_y0;
};

template <bool FAST>
__device__ __forceinline__ BinPartials2 rqs_partials_at(v2f z, v2f omz, v2f x, v2f bw, v2f bh, v2f d0, v2f d1,
                                                        float lo, float hi) {
  using M = Math<FAST>;
  BinPartials2 p;
  const v2f ibw = M::rcp(bw);
  const v2f s = bh * ibw;
  const v2f q = z * omz, qp = omz - z;
  const v2f st = d0 + d1 - s * 2.0f;
  const v2f Nn = vfma(s * z, z, d0 * q);
  const v2f den = vfma(st, q, s);
  const v2f iden = M::rcp(den);
  const v2f A = vfma(d1 * z, z, vfma(s * 2.0f, q, d0 * omz * omz));
  const v2f iA = M::rcp(A);
  const v2f iden2 = iden * iden;
  p.f_x = s * s * A * iden2;
  const v2f bq = bh * q * iden2;
  const v2f f_s = bq * vfma(Nn, splat<v2f>(2.0f), vfma(z * z, st, -d0));
  p.f_d0 = bq * omz * vfma(s, omz, d1 * z);
  p.f_d1 = -bq * Nn;
  p.f_x0 = -p.f_x;
  const v2f fsi = f_s * ibw;
  p.f_bw = -vfma(p.f_x, z, fsi * s);
  p.f_bh = vfma(Nn, iden, fsi);
  const v2f Az = (vfma(d1, z, s * qp) - d0 * omz) * 2.0f;
  const v2f denz = st * qp;
  const v2f l_z = vfma(denz * -2.0f, iden, Az * iA);
  const v2f l_s = q * 2.0f * vfma((d0 + d1) * iden, M::rcp(s), iA);
  p.l_x = l_z * ibw;
  p.l_x0 = -p.l_x;
  p.l_bw = -vfma(l_z, z, l_s * s) * ibw;
  p.l_bh = l_s * ibw;
  const v2f q2i = q * 2.0f * iden;
  p.l_d0 = vfma(omz * omz, iA, -q2i);
  p.l_d1 = vfma(z * z, iA, -q2i);
  p.f_y0 = splat<v2f>(1.0f);
  if (maybe_outside(x, lo, hi)) {      // (wave-level: rare) the linear tails hang on the fixed corners
    const v2i below = vle(x, lo), above = vge(x, hi), out = below | above;
    const v2f zero = splat<v2f>(0.0f);
    p.f_x = vsel(below, d0, vsel(above, d1, p.f_x));
    p.f_d0 = vsel(below, x - lo, vsel(above, zero, p.f_d0));
    p.f_d1 = vsel(above, x - hi, vsel(below, zero, p.f_d1));
    p.l_d0 = vsel(below, M::rcp(d0), vsel(above, zero, p.l_d0));
    p.l_d1 = vsel(above, M::rcp(d1), vsel(below, zero, p.l_d1));
    p.f_x0 = vsel(out, zero, p.f_x0); p.f_bw = vsel(out, zero, p.f_bw); p.f_bh = vsel(out, zero, p.f_bh);
    p.l_x = vsel(out, zero, p.l_x); p.l_x0 = vsel(out, zero, p.l_x0); p.l_bw = vsel(out, zero, p.l_bw);
    p.l_bh = vsel(out, zero, p.l_bh); p.f_y0 = vsel(out, zero, p.f_y0);
  }
  return p;
}

struct BinAdjoint2 { v2f v, x0, y0, bw, bh, d0, d1; };

template <bool INV, bool FAST>
__device__ __forceinline__ BinAdjoint2 bin_adjoint(const BinPartials2& p, v2f o_bar, v2f l_bar) {
  BinAdjoint2 a;
  if (!INV) {
    a.v = vfma(o_bar, p.f_x, l_bar * p.l_x);
    a.x0 = vfma(o_bar, p.f_x0, l_bar * p.l_x0);
    a.y0 = o_bar * p.f_y0;
    a.bw = vfma(o_bar, p.f_bw, l_bar * p.l_bw);
    a.bh = vfma(o_bar, p.f_bh, l_bar * p.l_bh);
    a.d0 = vfma(o_bar, p.f_d0, l_bar * p.l_d0);
    a.d1 = vfma(o_bar, p.f_d1, l_bar * p.l_d1);
  } else {
    a.v = vfma(-l_bar, p.l_x, o_bar) * Math<FAST>::rcp(p.f_x);
    a.x0 = -vfma(p.f_x0, a.v, l_bar * p.l_x0);
    a.y0 = -a.v * p.f_y0;
    a.bw = -vfma(p.f_bw, a.v, l_bar * p.l_bw);
    a.bh = -vfma(p.f_bh, a.v, l_bar * p.l_bh);
    a.d0 = -vfma(p.f_d0, a.v, l_bar * p.l_d0);
    a.d1 = -vfma(p.f_d1, a.v, l_bar * p.l_d1);
  }
  return a;
}

// the partials at the point the direction differentiates at: the input (INV = false), or the output of the inverse
// map, formed here from the bin's own quantities and from the nearer end of the bin (rqs_partials_inv above)
template <bool INV, bool FAST>
__device__ __forceinline__ BinPartials2 rqs_partials_dir(v2f v, v2f x0, v2f y0, v2f bw, v2f bh, v2f d0, v2f d1,
                                                         float lo, float hi) {
  using M = Math<FAST>;
  if constexpr (!INV) {
    const v2f z = clip01((v - x0) * M::rcp(bw));
    return rqs_partials_at<FAST>(z, 1.0f - z, v, bw, bh, d0, d1, lo, hi);
  } else {
    const v2f sl = bh * M::rcp(bw), st = d0 + d1 - sl * 2.0f;
    const v2f dy = vclamp(v - y0, bh), dyt = bh - dy;
    const v2i low = vle(dy - dyt, 0.0f);
    const v2f t = vsel(low, dy, dyt), da = vsel(low, d0, d1);
    const v2f c = -sl * t, b = vfma(-st, t, da * bh), a2 = vfma(sl, bh, -b);
    const v2f disc = vfma(b, b, a2 * c * -4.0f);
    const v2f r = clip01(M::div(c * -2.0f, b + M::sqrt(disc)));
    const v2f omr = 1.0f - r;
    const v2f z = vsel(low, r, omr), omz = vsel(low, omr, r);
    v2f out = vsel(low, vfma(bw, r, x0), vfma(-bw, r, x0 + bw));
    if (maybe_outside(v, lo, hi)) {
      out = vsel(vle(v, lo), vfma(v - lo, M::rcp(d0), splat<v2f>(lo)), out);
      out = vsel(vge(v, hi), vfma(v - hi, M::rcp(d1), splat<v2f>(hi)), out);
    }
    return rqs_partials_at<FAST>(z, omz, out, bw, bh, d0, d1, lo, hi);
  }
}

// 0 / 1 masks m[j] = [v > knot j] of a sample pair, j = 1 .. K-1 (bin_of_pairs' packed clamp-FMAs; `kb` = the F_XKB /
// F_YKB field of the prepared table); m[0] = 1, m[K] = 0: the one-hot of the bin is m[j] - m[j+1]
template <int K> __device__ __forceinline__ void bin_masks_pairs(const float* kb, v2f v, v2f (&m)[K + 1]) {
  const v2f big = v2f{1.152921504606846976e18f, 1.152921504606846976e18f};       // 2^60
  m[0] = splat<v2f>(1.0f); m[K] = splat<v2f>(0.0f);
#pragma unroll
  for (int j = 1; j < K; ++j) {
    const v2f nk = *reinterpret_cast<const v2f*>(kb + 2 * j);
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(m[j]) : "v"(v), "v"(big), "v"(nk));
  }
}

// table_spline_bwd for a sample pair: returns the adjoint of the input, accumulates the pair's bin adjoints into the
// lane's (pair-wide) per-bin sums
template <int K, bool INV, bool FAST>
__device__ __forceinline__ v2f table_spline_bwd(const float* tab, v2f v, v2f o_bar, v2f l_bar, const SplineConsts sc,
                                                v2f (&Wb)[K], v2f (&Hb)[K], v2f (&Db)[K + 1]) {
  v2f m[K + 1];
  bin_masks_pairs<K>(tab + tab_off(INV ? F_YKB : F_XKB, K), v, m);
  v2f kf = m[1];
#pragma unroll
  for (int j = 2; j < K; ++j) kf += m[j];
  const BinRow<v2f> row(tab, v2i{(int)kf.x, (int)kf.y});
  const v2f x0 = row.template get<K>(F_X0), y0 = row.template get<K>(F_Y0), bw = row.template get<K>(F_BW),
            bh = row.template get<K>(F_BH), d0 = row.template get<K>(F_D0), d1 = row.template get<K>(F_D1);
  const BinPartials2 p = rqs_partials_dir<INV, FAST>(v, x0, y0, bw, bh, d0, d1, sc.lo, sc.hi);
  const BinAdjoint2 a = bin_adjoint<INV, FAST>(p, o_bar, l_bar);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const v2f eq = m[j] - m[j + 1];                          // [k == j]
    Wb[j] = vfma(m[j + 1], a.x0, vfma(eq, a.bw, Wb[j]));     // knots left of the bin move with x0, the bin's own width with bw
    Hb[j] = vfma(m[j + 1], a.y0, vfma(eq, a.bh, Hb[j]));
    Db[j] = vfma(eq, a.d0, Db[j]);
    Db[j + 1] = vfma(eq, a.d1, Db[j + 1]);
  }
  return a.v;
}

// cond_spline_bwd for a sample pair fed from the conditioner tables.  qa / qb: the 2K softmax logits of sample a / b
// as K pairs (logit 2j, logit 2j + 1) in LOG2 units (pwl_logit_pairs); slopes(ka, kb, ta, tb) returns each sample's
// (t_k, t_k+1), log2 units with the softplus offset added (pwl_slope_pair).  Writes the adjoints of the 2K logits in
// NATURAL units to tb (sample pairs), the bin index kk (as floats) and the adjoints sb0 / sb1 of the bin's two slope
// logits; returns the adjoint of the spline input.  INV: the output is formed here, in the bin selected here.
template <int K, bool INV, bool FAST, class SlopeFetch>
__device__ __forceinline__ v2f cond_spline_bwd_rows(const v2f (&qa)[K], const v2f (&qb)[K], SlopeFetch&& slopes, v2f v,
                                                    v2f o_bar, v2f l_bar, const SplineConsts sc, v2f (&tb)[2 * K],
                                                    v2f& kk, v2f& sb0, v2f& sb1, bool shift_free = false) {
  using M = Math<FAST>;
  typedef v2f T;
  auto la = [&](int j) { return qa[j >> 1][j & 1]; };
  auto lb = [&](int j) { return qb[j >> 1][j & 1]; };
  T pw[K], ph[K];
  if (shift_free) {      // (wave-uniform) the table builder has checked the samples' cells: each group's maximum within +-4 of 0
#pragma unroll
    for (int k = 0; k < K; ++k) {
      pw[k] = v2f{__builtin_amdgcn_exp2f(la(k)), __builtin_amdgcn_exp2f(lb(k))};
      ph[k] = v2f{__builtin_amdgcn_exp2f(la(K + k)), __builtin_amdgcn_exp2f(lb(K + k))};
    }
  } else {
    float mwa = la(0), mwb = lb(0), mha = la(K), mhb = lb(K);
#pragma unroll
    for (int k = 1; k < K; ++k) {
      mwa = fmaxf(mwa, la(k)); mwb = fmaxf(mwb, lb(k));
      mha = fmaxf(mha, la(K + k)); mhb = fmaxf(mhb, lb(K + k));
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      pw[k] = v2f{__builtin_amdgcn_exp2f(la(k) - mwa), __builtin_amdgcn_exp2f(lb(k) - mwb)};
      ph[k] = v2f{__builtin_amdgcn_exp2f(la(K + k) - mha), __builtin_amdgcn_exp2f(lb(K + k) - mhb)};
    }
  }
  T sw = pw[0], sh = ph[0];
#pragma unroll
  for (int k = 1; k < K; ++k) { sw += pw[k]; sh += ph[k]; }
  const T isw = M::rcp(sw), ish = M::rcp(sh);
#pragma unroll
  for (int k = 0; k < K; ++k) { pw[k] *= isw; ph[k] *= ish; }      // softmax probabilities
  const T big = splat<T>(1.152921504606846976e18f);       // 2^60
  T px = splat<T>(sc.lo), py = splat<T>(sc.lo);            // running knot k
  T wp = vfma(pw[0], splat<T>(sc.span_eff), splat<T>(sc.min_bin)), hp = vfma(ph[0], splat<T>(sc.span_eff), splat<T>(sc.min_bin));
  T x0 = px, y0 = py;
  T m[K + 1];                                              // m[k] = [v > knot k]
  m[0] = splat<T>(1.0f); m[K] = splat<T>(0.0f);
  T bw = splat<T>(0.0f), bh = bw, cumw = bw, cumh = bw, pkw = bw, pkh = bw;
#pragma unroll
  for (int k = 1; k < K; ++k) {
    px += wp;
    py += hp;
    m[k] = step_mask(v - (INV ? py : px), big);
    const T o = m[k - 1] - m[k];                            // one-hot of bin k-1
    bw = k == 1 ? o * wp : vfma(o, wp, bw);
    bh = k == 1 ? o * hp : vfma(o, hp, bh);
    pkw = k == 1 ? o * pw[0] : vfma(o, pw[k - 1], pkw);
    pkh = k == 1 ? o * ph[0] : vfma(o, ph[k - 1], pkh);
    cumw = k == 1 ? m[k] * pw[0] : vfma(m[k], pw[k - 1], cumw);
    cumh = k == 1 ? m[k] * ph[0] : vfma(m[k], ph[k - 1], cumh);
    x0 = vfma(m[k], wp, x0);
    y0 = vfma(m[k], hp, y0);
    if (k == K - 1) { wp = sc.hi - px; hp = sc.hi - py; }   // last knot is exactly hi
    else { wp = vfma(pw[k], splat<T>(sc.span_eff), splat<T>(sc.min_bin)); hp = vfma(ph[k], splat<T>(sc.span_eff), splat<T>(sc.min_bin)); }
  }
  bw = vfma(m[K - 1], wp, bw);
  bh = vfma(m[K - 1], hp, bh);
  pkw = vfma(m[K - 1], pw[K - 1], pkw);
  pkh = vfma(m[K - 1], ph[K - 1], pkh);
  kk = m[1];
#pragma unroll
  for (int k = 2; k < K; ++k) kk += m[k];
  v2f ta, tbv;                                              // (t_k, t_k+1) of sample a, of sample b
  slopes((int)kk.x, (int)kk.y, ta, tbv);
  // natural units; the product is ROUNDED before |.| (see cond_spline_masked).  One exponential serves the softplus
  // and its derivative: e = exp(-|n|), d = relu(n) + log1p(e) + min_slope, sigmoid(n) = (n >= 0 ? 1 : e) / (1 + e)
  T n0 = v2f{ta.x, tbv.x} * LN2, n1 = v2f{ta.y, tbv.y} * LN2;
  asm volatile("" : "+v"(n0), "+v"(n1));
  T d0, d1, sg0, sg1;
  {
    const T a0 = vabs(n0), a1 = vabs(n1);
    const T e0 = M::exp(-a0), e1 = M::exp(-a1);
    const T ope0 = e0 + 1.0f, ope1 = e1 + 1.0f;
    const T l0 = vsel(vlt(e0, 1e-4f), vfma(e0 * -0.5f, e0, e0), M::log(ope0));
    const T l1 = vsel(vlt(e1, 1e-4f), vfma(e1 * -0.5f, e1, e1), M::log(ope1));
    d0 = vfma(n0 + a0, splat<T>(0.5f), l0) + sc.min_slope;
    d1 = vfma(n1 + a1, splat<T>(0.5f), l1) + sc.min_slope;
    const T i0 = M::rcp(ope0), i1 = M::rcp(ope1);
    sg0 = vsel(vge(n0, 0.0f), i0, e0 * i0);
    sg1 = vsel(vge(n1, 0.0f), i1, e1 * i1);
  }
  const BinPartials2 p = rqs_partials_dir<INV, FAST>(v, x0, y0, bw, bh, d0, d1, sc.lo, sc.hi);
  const BinAdjoint2 a = bin_adjoint<INV, FAST>(p, o_bar, l_bar);
  // widths: w_j = span p_j + min_bin, x0 = lo + sum_{j<k} w_j, bw = w_k:  theta_bar_j = span p_j (wbar_j - sum_i wbar_i p_i)
  const T Sw = vfma(a.x0, cumw, a.bw * pkw), Sh = vfma(a.y0, cumh, a.bh * pkh);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const T o = m[j] - m[j + 1];
    const T wb = vfma(m[j + 1], a.x0, vfma(o, a.bw, -Sw));
    const T hb = vfma(m[j + 1], a.y0, vfma(o, a.bh, -Sh));
    tb[j] = pw[j] * sc.span_eff * wb;
    tb[K + j] = ph[j] * sc.span_eff * hb;
  }
  sb0 = a.d0 * sg0; sb1 = a.d1 * sg1;
  return a.v;
}

// ---------------------------------------------------------------------------
// Conditioner forward that keeps the hidden activations (M = 2, H = P = 16).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void conditioner_keep(uniform_ptr w, int d, float c, const float* col, int first_idx,
                                                 int idx_step, int stride, float (&h1)[16], float (&h2)[16],
                                                 float (&th)[16]) {
  w = launder(w);
  uniform_ptr b0 = w + (1 + d) * 16;
  {
    float wc[16], bb[16];
    load_row<16>(w, wc); load_row<16>(b0, bb);
#pragma unroll
    for (int j = 0; j < 16; ++j) h1[j] = fmaf(wc[j], c, bb[j]);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int q = 0; q < d; ++q) {
    const float v = col[(first_idx + q * idx_step) * stride];
    float wr[16];
    load_row<16>(w + (1 + q) * 16, wr);
#pragma unroll
    for (int j = 0; j < 16; ++j) h1[j] = fmaf(wr[j], v, h1[j]);
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) h1[j] = fmaxf(h1[j], 0.0f);
  materialize<16>(h1);
  w = b0 + 16;
  {
    float bb[16];
    load_row<16>(w + 256, bb);
#pragma unroll
    for (int j = 0; j < 16; ++j) h2[j] = bb[j];
    __builtin_amdgcn_sched_barrier(0);
    dense_acc<16, 16, 1, float>(w, h1, h2);
#pragma unroll
    for (int j = 0; j < 16; ++j) h2[j] = fmaxf(h2[j], 0.0f);
    materialize<16>(h2);
  }
  w += 256 + 16;
  {
    float bb[16];
    load_row<16>(w + 256, bb);
#pragma unroll
    for (int j = 0; j < 16; ++j) th[j] = bb[j];
    __builtin_amdgcn_sched_barrier(0);
    dense_acc<16, 16, 1, float>(w, h2, th);
    materialize<16>(th);
  }
}

// out[i] = sum_j W[i][j] * in[j]  (data backprop through a layer y = x W)
__device__ __forceinline__ void dense_T(uniform_ptr W, const float (&in)[16], float (&out)[16]) {
#pragma unroll
  for (int i = 0; i < 16; i += 2) {
    float w0[16], w1[16];
    load_row<16>(W + i * 16, w0);
    load_row<16>(W + (i + 1) * 16, w1);
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { a0 = fmaf(w0[j], in[j], a0); a1 = fmaf(w1[j], in[j], a1); }
    out[i] = a0; out[i + 1] = a1;
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---------------------------------------------------------------------------
// dW[i][j] += sum over the wave's 64 samples of a_i * b_j, on the matrix core.
// `stage` is this wave's LDS staging region (16 x STG floats; the two operands pass through it in turn).  Sample s goes
// to MFMA step n = s & 15, k-slot g = s >> 4.  dW is row-major [rows][16] in
// this wave's private gradient slab; only rows < n_rows are stored.  If db is
// non-null the column sums of b are added to db[16].
// ---------------------------------------------------------------------------
constexpr int STG = 68;   // row stride of the staging tiles: conflict-free ds_read_b128

// The accumulator tile lives in the wave's gradient slab in global memory (L2): its read is a dependent ~1-2 us
// round trip if issued where it is needed, so conditioner_bwd issues the reads of all three tiles (and bias rows)
// up front (wgrad_fetch) and they arrive while the wave recomputes and back-propagates.
struct WgradAcc { f4 acc; float bias; };

__device__ __forceinline__ WgradAcc wgrad_fetch(const float* __restrict__ dW, int n_rows, const float* __restrict__ db) {
  const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
  WgradAcc r;
#pragma unroll
  for (int q = 0; q < 4; ++q) r.acc[q] = (4 * g + q < n_rows) ? dW[(4 * g + q) * 16 + i] : 0.0f;
  r.bias = (db && g == 0) ? db[i] : 0.0f;
  return r;
}

__device__ __forceinline__ void wgrad_mfma(float* stage, const float (&a)[16], const float (&b)[16],
                                           float* __restrict__ dW, int n_rows, float* __restrict__ db, WgradAcc pre) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, i = lane & 15;
  f4 av[4], bv[4];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < 16; ++r) stage[r * STG + lane] = a[r];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int n = 0; n < 4; ++n) av[n] = *reinterpret_cast<const f4*>(stage + i * STG + 16 * g + 4 * n);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < 16; ++r) stage[r * STG + lane] = b[r];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int n = 0; n < 4; ++n) bv[n] = *reinterpret_cast<const f4*>(stage + i * STG + 16 * g + 4 * n);
  __builtin_amdgcn_wave_barrier();
  f4 acc = pre.acc;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[n][e], bv[n][e], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) if (4 * g + r < n_rows) dW[(4 * g + r) * 16 + i] = acc[r];
  if (db) {
    float sum = 0.0f;
#pragma unroll
    for (int n = 0; n < 4; ++n) sum += (bv[n][0] + bv[n][1]) + (bv[n][2] + bv[n][3]);
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    if (g == 0) db[i] = pre.bias + sum;
  }
}

// the three accumulator tiles of one conditioner (output, hidden, first layer), fetched before the wave starts
// recomputing the conditioner: ~1 500 instructions of cover for the L2 round trip
struct WgradPre { WgradAcc o, h, f; };
__device__ __forceinline__ WgradPre wgrad_prefetch(const float* __restrict__ gw, int d) {
  const int o_b0 = (1 + d) * 16, o_w1 = o_b0 + 16, o_b1 = o_w1 + 256, o_wo = o_b1 + 16, o_bo = o_wo + 256;
  WgradPre p;
  p.o = wgrad_fetch(gw + o_wo, 16, gw + o_bo);
  p.h = wgrad_fetch(gw + o_w1, 16, gw + o_b1);
  p.f = wgrad_fetch(gw, d + 2, nullptr);
  // keep the loads here (the scheduler would sink them to their first use) -- a scheduling barrier, not a use of
  // the values: a use would make the wave wait for the loads on the spot
  __builtin_amdgcn_sched_barrier(0);
  return p;
}

// ---------------------------------------------------------------------------
// The backward pass's conditioner work on the matrix cores (hidden 16, P = 16, one sample per lane, 64 samples =
// 4 groups of 16 per wave).  EVERY matrix product of a conditioner -- the three layers of the (re)computation, the
// three data-backprop products and the three weight-gradient GEMMs -- is a run of v_mfma_f32_16x16x4_f32, and no
// loop over the d conditioning inputs is left on the vector ALU (round 2 evaluated the first layer, its weight
// gradient's operand and the input adjoints per lane in runtime loops over d: scalar weight loads and LDS round trips
// waited for one after the other, ~1 500 vector instructions per conditioner at dim 10).
//
// Operand layouts (lane l: g = l >> 4, i = s = l & 15; checked on hardware by scripts/probes/mfma_probe.hip):
//   A lane (g, i) holds A[i][k = g], B lane (g, s) holds B[k = g][n = s], result lane (g, s) register r holds
//   D[row 4g + r][col s].  Hidden layers permute k to 4g + t (step t) so that a layer's result registers ARE the next
//   layer's B operands ("MFMA layout": lane (g, s), register r of group q: unit 4g + r of sample 16q + s).
// First layer: the inputs [c, v_1 .. v_d, 1] (the constant row carries the bias) are ROWS of the tile's LDS buffers
// (a row = one input dimension, 64 consecutive samples per wave), so the B operand of step t, k = 4t + g, is one
// ds_read_b32 per group at row k, and steps beyond ceil((d + 2) / 4) are skipped; the A operand of that step is the
// flat weight block itself, W0pad[k][i] = wflat[16 k + i] (rows 0 .. d: W0, row d + 1: b0 -- contiguous in the
// flat layout).  The same rows, read 16 bytes per lane along the samples, are the A operand of the first layer's
// weight-gradient GEMM (K = samples): rows are GTS + 4 floats apart, so the 16 lanes of a group hit 64 different banks.
// ---------------------------------------------------------------------------
struct CondGeom {
  int in_off;       // LDS offset (floats) of row 0 of the buffer that holds the conditioning inputs, at the wave's first sample
  int c_off;        // ... of the wave's 64 conditions
  int ones_off;     // ... of 64 x 1.0f (shared by the workgroup)
  int first_idx, idx_step, stride, d;
};

// LDS offset of input row k of a conditioner: k = 0 the condition, 1 .. d the conditioning inputs, d + 1 (and beyond:
// the callers discard or zero what comes of those rows) ones
__device__ __forceinline__ int cond_row_off(const CondGeom& G, int k) {
  int kk = k < 1 ? 1 : k;
  kk = kk > G.d ? G.d : kk;
  int off = G.in_off + (G.first_idx + (kk - 1) * G.idx_step) * G.stride;
  off = k == 0 ? G.c_off : off;
  off = k > G.d ? G.ones_off : off;
  return off;
}

// The weights of one conditioner's forward evaluation in MFMA operand form: fetched (global memory, L2) one
// conditioner AHEAD of their use -- a dependent L2 round trip costs a lone wave 1-2 us, as long as the conditioner's
// arithmetic.  a0[t]: W0pad[4t + g][i] (0 beyond row d + 1); A1 / A2, bias1 / bias2: the 16 x 16 layers (prepare_kernel's `wq`).
// (the output layer's A2 / bias2 are fetched at the start of the evaluation itself: the first two layers cover them)
struct CondW { float a0[4]; f4 A1, bias1; int q2; };      // q2: (wave-uniform) offset of the output layer's operands in `wq`, in f4 units

// wq_layer: the flow layer's MFMA-layout weights, q_off: this conditioner's offset in them (floats)
__device__ __forceinline__ CondW cond_weights(const float* __restrict__ wq_layer, int q_off, const float* __restrict__ wflat, int d) {
  const int lane = threadIdx.x & 63, g = lane >> 4, s = lane & 15;
  const int q1 = (q_off >> 2) + (2 + d) * 64;
  const f4* p = reinterpret_cast<const f4*>(wq_layer) + q1;
  CondW w;
#pragma unroll
  for (int t = 0; t < 4; ++t) w.a0[t] = 4 * t + g <= d + 1 ? wflat[(4 * t + g) * 16 + s] : 0.0f;
  w.A1 = p[lane]; w.bias1 = p[64 + lane]; w.q2 = q1 + 128;
  return w;
}

// MFMA-layout values -> a staging region ([unit][sample], row stride STG): unit 4g + t of sample 16q + i
__device__ __forceinline__ void stage_m(float* region, const float (&m)[4][4]) {
  const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int t = 0; t < 4; ++t) region[(4 * g + t) * STG + 16 * q + i] = m[q][t];
  }
}

constexpr int STAGE_FLOATS = 16 * STG;           // per wave: ONE [unit][sample] region, the two operands of a weight-gradient GEMM pass through it in turn

// Conditioner forward (M = 2) on the matrix cores: theta in the lane layout; both hidden activations are handed back
// in MFMA layout (the backward's ReLU masks are their signs, and they are the operands of its weight-gradient GEMMs).
__device__ __forceinline__ void cond_fwd_mfma(const float* lds, const CondGeom& G, const float* __restrict__ wq_layer, const CondW& w,
                                              float (&h1m)[4][4], float (&h2m)[4][4], float (&th)[16]) {
  const int lane = threadIdx.x & 63, g = lane >> 4, s = lane & 15;
  const int nst = (G.d + 5) >> 2;                  // k-steps of the first layer: ceil((d + 2) / 4)
  const f4* p2 = reinterpret_cast<const f4*>(wq_layer) + w.q2 + lane;
  const f4 A2 = p2[0], bias2 = p2[64];
  __builtin_amdgcn_sched_barrier(0);
  f4 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t < nst) {                                 // (wave-uniform)
      const float* row = lds + cond_row_off(G, 4 * t + g) + s;
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.a0[t], row[16 * q], acc[q], 0, 0, 0);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int t = 0; t < 4; ++t) h1m[q][t] = fmaxf(acc[q][t], 0.0f);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = w.bias1;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.A1[t], h1m[q][t], acc[q], 0, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int t = 0; t < 4; ++t) h2m[q][t] = fmaxf(acc[q][t], 0.0f);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = bias2;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[t], h2m[q][t], acc[q], 0, 0, 0);
  }
  float thm[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int r = 0; r < 4; ++r) thm[q][r] = acc[q][r];
  }
  from_mfma_layout(thm, th);
}

// The A operands of a conditioner's three data-backprop products, out_i = sum_j W[i][j] in_j with j = 4g + t: one
// 16-byte load each from the flat weights.  The output layer's is issued before the spline backward (which covers the
// round trip), the other two at the start of the conditioner backward (covered by its first stage).
__device__ __forceinline__ f4 cond_weight_T(const float* __restrict__ wmat) {
  const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
  return *reinterpret_cast<const f4*>(wmat + i * 16 + 4 * g);
}

// One operand of a weight-gradient GEMM (K = the wave's 64 samples): MFMA-layout values -> the staging region ->
// the lane's 4 x 16 bytes along the samples of unit i (A and B operands have the same form: lane (g, i), k = sample)
struct WgradOp { f4 v[4]; };
__device__ __forceinline__ WgradOp wgrad_operand(float* region, const float (&m)[4][4]) {
  const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
  __builtin_amdgcn_wave_barrier();
  stage_m(region, m);
  __builtin_amdgcn_wave_barrier();
  WgradOp o;
#pragma unroll
  for (int n = 0; n < 4; ++n) o.v[n] = *reinterpret_cast<const f4*>(region + i * STG + 16 * g + 4 * n);
  __builtin_amdgcn_wave_barrier();
  return o;
}
// ... or straight from the input rows of the conditioner (row k = lane's i; the ones row yields the bias gradient as row d + 1)
__device__ __forceinline__ WgradOp wgrad_operand_inputs(const float* lds, const CondGeom& G) {
  const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
  const float* ra = lds + cond_row_off(G, i) + 16 * g;
  WgradOp o;
#pragma unroll
  for (int n = 0; n < 4; ++n) o.v[n] = *reinterpret_cast<const f4*>(ra + 4 * n);
  return o;
}

// dW[rows < n_rows][16] += a b^T over the wave's 64 samples; db (optional) += the column sums of b
__device__ __forceinline__ void wgrad_apply(const WgradOp& a, const WgradOp& b, float* __restrict__ dW, int n_rows,
                                            float* __restrict__ db, WgradAcc pre) {
  const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
  f4 acc = pre.acc;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[n][e], b.v[n][e], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) if (4 * g + r < n_rows) dW[(4 * g + r) * 16 + i] = acc[r];
  if (db) {
    float sum = 0.0f;
#pragma unroll
    for (int n = 0; n < 4; ++n) sum += (b.v[n][0] + b.v[n][1]) + (b.v[n][2] + b.v[n][3]);
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    if (g == 0) db[i] = pre.bias + sum;
  }
}

// Conditioner backward on the matrix cores.  h1m / h2m: the hidden activations of cond_fwd_mfma, tb: theta_bar in the
// lane layout, adj_off: LDS offset (wave's first sample, row 0) of the buffer that receives the adjoints of the
// conditioning inputs, gw: this conditioner's block of the wave's gradient slab.  Weight gradients: h2 x theta_bar -> dWo,
// h1 x g2 -> dW1, (input rows) x g1 -> dW0, each operand through the wave's one staging region.
template <bool WGRAD = true>
__device__ __forceinline__ void cond_bwd_mfma(float* lds, const CondGeom& G, int adj_off, const float* __restrict__ wflat,
                                              f4 Ao, const float (&h1m)[4][4], const float (&h2m)[4][4],
                                              const float (&tb)[16], float* __restrict__ gw, float* stage, WgradAcc pre_o) {
  const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
  const int d = G.d;
  const int o_b0 = (1 + d) * 16, o_w1 = o_b0 + 16, o_b1 = o_w1 + 256, o_wo = o_b1 + 16, o_bo = o_wo + 256;
  const f4 A1 = cond_weight_T(wflat + o_w1), A0 = cond_weight_T(wflat);       // (A0: rows beyond d give results nobody reads)
  [[maybe_unused]] WgradAcc pre_h, pre_f;
  if constexpr (WGRAD) { pre_h = wgrad_fetch(gw + o_w1, 16, gw + o_b1); pre_f = wgrad_fetch(gw, d + 2, nullptr); }
  __builtin_amdgcn_sched_barrier(0);
  float tbm[4][4];
  to_mfma_layout(tb, tbm);
  if constexpr (WGRAD) {
    const WgradOp oa = wgrad_operand(stage, h2m), ob = wgrad_operand(stage, tbm);
    wgrad_apply(oa, ob, gw + o_wo, 16, gw + o_bo, pre_o);
  }
  float g2m[4][4];
  {
    f4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ao[t], tbm[q][t], acc[q], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int r = 0; r < 4; ++r) g2m[q][r] = h2m[q][r] > 0.0f ? acc[q][r] : 0.0f;
    }
  }
  if constexpr (WGRAD) {
    const WgradOp oa = wgrad_operand(stage, h1m), ob = wgrad_operand(stage, g2m);
    wgrad_apply(oa, ob, gw + o_w1, 16, gw + o_b1, pre_h);
  }
  float g1m[4][4];
  {
    f4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[t], g2m[q][t], acc[q], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int r = 0; r < 4; ++r) g1m[q][r] = h1m[q][r] > 0.0f ? acc[q][r] : 0.0f;
    }
  }
  if constexpr (WGRAD) {
    const WgradOp oa = wgrad_operand_inputs(lds, G), ob = wgrad_operand(stage, g1m);
    wgrad_apply(oa, ob, gw, d + 2, nullptr, pre_f);
  }
  // adjoints of the conditioning inputs: row k = 4g + r of W0 g1 belongs to input k (1 .. d); every (input, sample)
  // has exactly one owner lane-register, so the accumulation is a plain read-modify-write
  {
    f4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[t], g1m[q][t], acc[q], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 4 * g + r;
      if (k >= 1 && k <= d) {
        float* row = lds + adj_off + (G.first_idx + (k - 1) * G.idx_step) * G.stride + i;
#pragma unroll
        for (int q = 0; q < 4; ++q) row[16 * q] += acc[q][r];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Conditioner backward (M = 2).  theta_bar -> weight gradients (into the
// wave's slab `gw`, laid out like the flat parameters of this conditioner) and
// adjoints of the d conditioning inputs (accumulated into adj_col).
// Lanes whose sample is beyond the batch must pass theta_bar = 0.
// ---------------------------------------------------------------------------
// WGRAD=false: only the adjoints of the conditioning inputs (vector-Jacobian
// products w.r.t. the points: forward_jac / inverse_jac / gauge_potential).
template <bool WGRAD = true>
__device__ __forceinline__ void conditioner_bwd(uniform_ptr w, int d, float c, const float* col, int first_idx,
                                                int idx_step, int stride, const float (&h1)[16],
                                                const float (&h2)[16], const float (&tb)[16], float* adj_col,
                                                float* __restrict__ gw, float* stage, const WgradPre& pre) {
  w = launder(w);
  const int o_b0 = (1 + d) * 16, o_w1 = o_b0 + 16, o_b1 = o_w1 + 256, o_wo = o_b1 + 16, o_bo = o_wo + 256;
  [[maybe_unused]] const WgradAcc pre_o = pre.o, pre_1 = pre.h, pre_0 = pre.f;
  // output layer
  if constexpr (WGRAD) wgrad_mfma(stage, h2, tb, gw + o_wo, 16, gw + o_bo, pre_o);
  float g2[16];
  dense_T(w + o_wo, tb, g2);
#pragma unroll
  for (int i = 0; i < 16; ++i) g2[i] = h2[i] > 0.0f ? g2[i] : 0.0f;
  materialize<16>(g2);
  // hidden layer
  if constexpr (WGRAD) wgrad_mfma(stage, h1, g2, gw + o_w1, 16, gw + o_b1, pre_1);
  float g1[16];
  dense_T(w + o_w1, g2, g1);
#pragma unroll
  for (int i = 0; i < 16; ++i) g1[i] = h1[i] > 0.0f ? g1[i] : 0.0f;
  materialize<16>(g1);
  // first layer: inputs [c, v_0..v_{d-1}, 1] (the constant row yields the bias gradient)
  if constexpr (WGRAD) {
    float in[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) in[r] = 0.0f;
    in[0] = c;
    for (int q = 0; q < d && q < 14; ++q) {
      const float v = col[(first_idx + q * idx_step) * stride];
#pragma unroll
      for (int r = 1; r < 15; ++r) in[r] = (r == q + 1) ? v : in[r];
    }
#pragma unroll
    for (int r = 1; r < 16; ++r) in[r] = (r == d + 1) ? 1.0f : in[r];
    wgrad_mfma(stage, in, g1, gw, d + 2, nullptr, pre_0);
  }
  for (int q = 0; q < d; ++q) {          // adjoints of the conditioning inputs
    float wr[16];
    load_row<16>(w + (1 + q) * 16, wr);
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = fmaf(wr[j], g1[j], acc);
    adj_col[(first_idx + q * idx_step) * stride] += acc;
  }
}

}  // namespace cnf
