// cnf_pwl.h -- exact piecewise-linear form of the conditioner at dim 2.
//
// At D = 2 the conditioner input is [c, u] with ONE per-sample scalar u, and
// every reference call site uses a condition that is uniform over the batch
// (per time-slice here): theta(u) = W_out^T relu(W_1^T relu(a u + b(c)) + b_1)
// + b_out is then a continuous piecewise-LINEAR function of u -- 16 first-layer
// breakpoints, at most one more per second-layer unit in each of the 17
// intervals: <= 1 + 16 + 17*16 = 289 pieces.  pwl_build_kernel computes, per
// (slice, flow layer), the sorted breakpoints and each piece's affine map
// u -> theta (16 slopes + 16 intercepts) in float64; the flow kernel then
// replaces the 2 -> 16 -> 16 -> 16 MLP (544 FMAs per sample) by a search in
// the breakpoints and 16 FMAs.  The function evaluated is the same network;
// only the rounding differs (the table is rounded once from float64).  Each
// piece's map is stored about a reference point near the samples it serves,
// theta = S (u - u_ref) + T with u_ref the point of the piece nearest to 0
// (clamped to the search grid): an intercept referred to u = 0 would cancel
// against S u for pieces far from the origin, one referred to the midpoint of a
// very wide piece would do the same for samples near its inner end.
#pragma once

#include "cnf_common.h"

namespace cnf {

constexpr int PWL_H = 16;                 // hidden width this path is built for
constexpr int PWL_P = 16;                 // spline parameters (K = 5)
constexpr int PWL_NBP = 320;              // sorted breakpoints, +inf padded (>= 289 + sentinel)
constexpr int PWL_NG = 512;               // coarse grid cells over [PWL_GMIN, PWL_GMAX)
constexpr int PWL_NPIECE = 289;
constexpr float PWL_GMIN = -16.0f;
constexpr float PWL_GSCALE = 16.0f;       // cells per unit
constexpr int PWL_NREF = 304;             // per-piece reference points (289, padded to 16 bytes)
constexpr int PWL_OFF_GRID = PWL_NBP;
constexpr int PWL_OFF_REF = PWL_NBP + PWL_NG;
constexpr int PWL_OFF_PIECE = PWL_NBP + PWL_NG + PWL_NREF;
// A piece's row: 16 slopes + 16 intercepts, padded to 36 floats.  Lanes gather rows at unrelated p with
// ds_read_b128; at a stride of 144 B consecutive rows start 4 x (9 p mod 16) banks apart, so the 64 lanes
// spread over all LDS banks, and every chunk of a row is an immediate offset from ONE address per sample
// (a 128-byte stride would put every lane on the same 8 banks; an XOR swizzle fixes that too but costs an
// address computation per chunk).
constexpr int PWL_ROW = 36;
constexpr int PWL_TBL = PWL_OFF_PIECE + PWL_NPIECE * PWL_ROW;          // floats per (slice, layer): 11 540
// The flow and loss kernels stage the header arrays and the first PWL_LROWS rows in LDS (networks met in
// practice have 30-50 pieces; 289 is the worst case); rows beyond that are read from the global table.
// 23 KB per table instead of 46 KB: the loss kernel keeps up to three table sets of L = 2 layers in LDS.
constexpr int PWL_LROWS = 128;
constexpr int PWL_LTBL = PWL_OFF_PIECE + PWL_LROWS * PWL_ROW;
constexpr int PWL_N_SLOT = PWL_NBP - 1;   // the piece count n, stored (as int bits) in the last padding slot of bp[]


// One block (512 threads) per (slice, layer).  Rows are written pre-scaled for the spline that
// consumes them (cond_spline_masked): the 2K softmax logits in log2 units (x log2 e), the K + 1 slope
// logits with the softplus offset added.  Rows past
// the last piece are never read (the search stops at the +inf padding) and are left unwritten.
// `c_offset` is added to the slice's condition (the loss kernels need t - dt/2 and t + dt/2).
constexpr int PWL_CHUNK = 64;             // pieces per pass of the two-stage affine-map computation

__global__ __launch_bounds__(512) void pwl_build_kernel(const float* __restrict__ weights /* prep + hdr */,
                                                        int64_t per_layer, const float* __restrict__ cvals,
                                                        float c_offset, int L, double sp_offset,
                                                        float* __restrict__ tables) {
  __shared__ double a1[PWL_H], b1v[PWL_H], W1[PWL_H * PWL_H], bb1[PWL_H], Wo[PWL_H * PWL_P], bo[PWL_P];
  __shared__ double bpu[PWL_H], sbp[PWL_H];
  __shared__ double candu[PWL_NPIECE - 1], cand[PWL_NPIECE - 1];
  __shared__ double PQ[PWL_CHUNK * PWL_H * 2];
  const int tid = threadIdx.x;
  const int slice = blockIdx.x / L, l = blockIdx.x % L;
  const float* w = weights + l * per_layer;         // D = 2: the layer's only conditioner (d = 1)
  const double c = (double)cvals[slice] + (double)c_offset;
  const double INF = __longlong_as_double(0x7ff0000000000000LL);
  // flat layout: W0[2][16] | b0[16] | W1[16][16] | b1[16] | Wout[16][16] | bout[16]
  if (tid < PWL_H) {
    const double a = (double)w[PWL_H + tid], b = (double)w[tid] * c + (double)w[2 * PWL_H + tid];
    a1[tid] = a;
    b1v[tid] = b;
    bpu[tid] = a != 0.0 ? -b / a : INF;
    bb1[tid] = (double)w[3 * PWL_H + 256 + tid];
    // spline-side pre-scaling of the output layer (see above)
    bo[tid] = (double)w[3 * PWL_H + 256 + PWL_H + 256 + tid];
  }
  for (int i = tid; i < 256; i += blockDim.x) {
    W1[i] = (double)w[3 * PWL_H + i];
    Wo[i] = (double)w[3 * PWL_H + 256 + PWL_H + i];
  }
  __syncthreads();
  if (tid < PWL_H) {         // rank sort of the 16 first-layer breakpoints
    const double v = bpu[tid];
    int r = 0;
    for (int j = 0; j < PWL_H; ++j) r += (bpu[j] < v || (bpu[j] == v && j < tid)) ? 1 : 0;
    sbp[r] = v;
  }
  __syncthreads();
  auto test_point = [&](double lo, double hi) -> double {
    const bool fl = lo > -INF, fh = hi < INF;
    return fl && fh ? 0.5 * (lo + hi) : (fl ? lo + 1.0 : (fh ? hi - 1.0 : 0.0));
  };
  // second-layer zero crossings inside each of the 17 first-layer intervals
  if (tid < PWL_H) candu[tid] = sbp[tid];
  if (tid < 17 * PWL_H) {
    const int i = tid / PWL_H, k = tid % PWL_H;
    const double lo = i == 0 ? -INF : sbp[i - 1], hi = i == PWL_H ? INF : sbp[i];
    double root = INF;
    if (lo < hi && lo < INF) {
      const double u = test_point(lo, hi);
      double P = 0.0, Q = bb1[k];
      for (int j = 0; j < PWL_H; ++j) {
        const double on = a1[j] * u + b1v[j] > 0.0 ? W1[j * PWL_H + k] : 0.0;
        P += on * a1[j];
        Q += on * b1v[j];
      }
      if (P != 0.0) { const double r = -Q / P; if (r > lo && r < hi) root = r; }
    }
    candu[PWL_H + tid] = root;
  }
  __syncthreads();
  // rank sort of the 288 candidates (+inf = none), ties by index
  double mine = INF;
  if (tid < PWL_NPIECE - 1) {
    mine = candu[tid];
    int r = 0;
    for (int j = 0; j < PWL_NPIECE - 1; ++j) { const double o = candu[j]; r += (o < mine || (o == mine && j < tid)) ? 1 : 0; }
    cand[r] = mine;
  }
  const int n = __syncthreads_count(mine < INF);      // finite breakpoints; pieces 0 .. n
  float* T = tables + (int64_t)blockIdx.x * PWL_TBL;
  for (int p = tid; p < PWL_NBP; p += blockDim.x)
    T[p] = p == PWL_N_SLOT ? __int_as_float(n) : (p < n ? (float)cand[p] : __int_as_float(0x7f800000));
  // coarse grid: number of breakpoints <= the cell's left edge (a lower bound for the scan).  The edge is
  // pulled in by 1e-4: pwl_cell() computes the cell in fp32, and u a rounding error below an edge may land
  // in the cell above it.
  for (int g = tid; g < PWL_NG; g += blockDim.x) {
    const double x = (double)PWL_GMIN + (double)g / (double)PWL_GSCALE - 1e-4;
    int lo_ = 0, hi_ = g == 0 ? 0 : n;          // cell 0 also serves every u below the grid: scan from piece 0
    while (lo_ < hi_) { const int mid = (lo_ + hi_) >> 1; if (cand[mid] <= x) lo_ = mid + 1; else hi_ = mid; }
    reinterpret_cast<int*>(T + PWL_OFF_GRID)[g] = lo_;
  }
  // affine map of every piece, PWL_CHUNK pieces per pass:
  //   stage 1, task (p, k): second-layer pre-activation P u + Q on the piece (zeroed if its ReLU is off)
  //   stage 2, task (p, m): theta_m = S u + T
  const double LOG2E_D = 1.4426950408889634;
  for (int base = 0; base <= n; base += PWL_CHUNK) {
    const int np = n + 1 - base < PWL_CHUNK ? n + 1 - base : PWL_CHUNK;
    for (int t = tid; t < np * PWL_H; t += blockDim.x) {
      const int p = base + (t >> 4), k = t & 15;
      const double lo = p == 0 ? -INF : cand[p - 1], hi = p < n ? cand[p] : INF;
      const double u = test_point(lo, hi);
      double P = 0.0, Q = bb1[k];
      for (int j = 0; j < PWL_H; ++j) {
        const double on = a1[j] * u + b1v[j] > 0.0 ? W1[j * PWL_H + k] : 0.0;
        P += on * a1[j];
        Q += on * b1v[j];
      }
      const bool act = P * u + Q > 0.0;
      PQ[2 * t] = act ? P : 0.0;
      PQ[2 * t + 1] = act ? Q : 0.0;
    }
    __syncthreads();
    for (int t = tid; t < np * PWL_P; t += blockDim.x) {
      const int pl = t >> 4, m = t & 15, p = base + pl;
      double S = 0.0, Tt = bo[m];
      for (int k = 0; k < PWL_H; ++k) {
        const double wo = Wo[k * PWL_P + m];
        S += wo * PQ[2 * (pl * PWL_H + k)];
        Tt += wo * PQ[2 * (pl * PWL_H + k) + 1];
      }
      // Refer the map to the point of the piece nearest to 0, kept inside the search grid: samples live
      // there, and a piece can be thousands wide (a midpoint reference at u ~ 1000 makes S (u - u_ref) and
      // T cancel catastrophically for u ~ 10).  The reference need not lie inside the piece -- T is the
      // value of the piece's affine map at u_ref, not of the network.
      const double lo = p == 0 ? -INF : cand[p - 1], hi = p < n ? cand[p] : INF;
      const double nearest = lo > 0.0 ? lo : (hi < 0.0 ? hi : 0.0);
      const float uref = (float)(nearest < (double)PWL_GMIN ? (double)PWL_GMIN : (nearest > -(double)PWL_GMIN ? -(double)PWL_GMIN : nearest));
      Tt += S * (double)uref;
      if (m == 0) T[PWL_OFF_REF + p] = uref;
      if (m < 10) { S *= LOG2E_D; Tt *= LOG2E_D; } else { Tt += sp_offset; }
      float* row = T + PWL_OFF_PIECE + p * PWL_ROW;
      row[m] = (float)S;
      row[PWL_P + m] = (float)Tt;
    }
    __syncthreads();
  }
}

// The compiler would pair the two samples' FMAs into v_pk_fma_f32 and pay ~40 v_mov to interleave
// the two gathered rows; a plain v_fma_f32 per half needs none.
__device__ __forceinline__ float fma_scalar(float a, float b, float c) {
  float r;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

typedef const f4 __attribute__((address_space(3))) * lds_f4_ptr;

__device__ __forceinline__ int pwl_cell(float u) {
  // truncation == floor for the cells that survive the clamp (negative values go to cell 0 either way)
  const int cell = (int)fmaf(u, PWL_GSCALE, -PWL_GMIN * PWL_GSCALE);
  return cell < 0 ? 0 : (cell > PWL_NG - 1 ? PWL_NG - 1 : cell);
}

// theta = S (u - u_ref) + T from row p (`gtbl`: the same table in global memory, for rows past the LDS window)
__device__ __forceinline__ void pwl_row(const float* tbl, const float* __restrict__ gtbl, int p, float u,
                                        float (&th)[PWL_P]) {
  const float du = u - tbl[PWL_OFF_REF + p];
  if (p < PWL_LROWS) {
    // ONE address per sample; the 8 chunks are immediate offsets of the ds_read_b128s
    const lds_f4_ptr row = (lds_f4_ptr)(uintptr_t)((uint32_t)(uintptr_t)(tbl + PWL_OFF_PIECE) + (uint32_t)p * (PWL_ROW * 4));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f4 s = row[q], t = row[4 + q];
#pragma unroll
      for (int e = 0; e < 4; ++e) th[4 * q + e] = fma_scalar(s[e], du, t[e]);
    }
  } else {
    const f4* row = reinterpret_cast<const f4*>(gtbl + PWL_OFF_PIECE + p * PWL_ROW);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f4 s = row[q], t = row[4 + q];
#pragma unroll
      for (int e = 0; e < 4; ++e) th[4 * q + e] = fma_scalar(s[e], du, t[e]);
    }
  }
}

__device__ __forceinline__ void pwl_eval(const float* tbl, const float* __restrict__ gtbl, float u, float (&th)[PWL_P]) {
  int p = reinterpret_cast<const int*>(tbl + PWL_OFF_GRID)[pwl_cell(u)];
  const float us = fminf(u, 3.0e38f);            // u = +inf must stop at the +inf padding too
  while (tbl[p] <= us) ++p;                      // bp[>= n] = +inf: terminates
  pwl_row(tbl, gtbl, p, u, th);
}

// Two samples: both searches advance in ONE loop (half the chain of dependent LDS reads).
__device__ __forceinline__ void pwl_eval(const float* tbl, const float* __restrict__ gtbl, v2f u, v2f (&th)[PWL_P]) {
  const int* grid = reinterpret_cast<const int*>(tbl + PWL_OFF_GRID);
  int px = grid[pwl_cell(u.x)], py = grid[pwl_cell(u.y)];
  const float ux = fminf(u.x, 3.0e38f), uy = fminf(u.y, 3.0e38f);
  bool more;
  do {
    const bool mx = tbl[px] <= ux, my = tbl[py] <= uy;
    px += mx ? 1 : 0;
    py += my ? 1 : 0;
    more = mx || my;
  } while (more);
  float tx[PWL_P], ty[PWL_P];
  pwl_row(tbl, gtbl, px, u.x, tx);
  pwl_row(tbl, gtbl, py, u.y, ty);
#pragma unroll
  for (int m = 0; m < PWL_P; ++m) th[m] = v2f{tx[m], ty[m]};
}

}  // namespace cnf
