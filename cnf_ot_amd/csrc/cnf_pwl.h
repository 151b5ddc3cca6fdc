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
constexpr int PWL_NBP = 320;              // sorted breakpoints, NaN padded (>= 289 + sentinel)
constexpr int PWL_NG = 2048;              // search grid cells over [PWL_GMIN, -PWL_GMIN), one uint16 each
constexpr int PWL_NPIECE = 289;
constexpr float PWL_GMIN = -16.0f;
constexpr float PWL_GSCALE = 64.0f;       // cells per unit
// grid entry: bits 0-8 the number of breakpoints <= the cell's left edge (a lower bound of the piece index),
// bit 14: more than two breakpoints can lie below a sample of this cell (the two unrolled comparisons do not
// finish the search), bit 15: the cell touches a piece that needs the general spline evaluation.
constexpr uint32_t PWL_G_INDEX = 0x1ff, PWL_G_MANY = 0x4000, PWL_G_GENERAL = 0x8000;
constexpr int PWL_NREF = 304;             // per-piece reference points (289, padded to 16 bytes)
constexpr int PWL_OFF_GRID = PWL_NBP;
constexpr int PWL_OFF_REF = PWL_NBP + PWL_NG / 2;
constexpr int PWL_OFF_PIECE = PWL_NBP + PWL_NG / 2 + PWL_NREF;
static_assert(PWL_OFF_PIECE % 4 == 0, "rows are read with 16-byte LDS loads");
// A piece's row: 16 slopes + 16 intercepts, padded to 36 floats.  Lanes gather rows at unrelated p with
// ds_read_b128; at a stride of 144 B consecutive rows start 4 x (9 p mod 16) banks apart, so the 64 lanes
// spread over all LDS banks, and every chunk of a row is an immediate offset from ONE address per sample
// (a 128-byte stride would put every lane on the same 8 banks; an XOR swizzle fixes that too but costs an
// address computation per chunk).
constexpr int PWL_ROW = 36;
constexpr int PWL_TBL = PWL_OFF_PIECE + PWL_NPIECE * PWL_ROW;          // floats per (slice, layer): 11 540
// The flow and loss kernels stage the header arrays and the first PWL_LROWS rows in LDS (networks met in
// practice have 30-50 pieces; 289 is the worst case); rows beyond that are read from the global table.
// 22 KB per table instead of 48 KB: the loss kernel keeps up to three table sets of L = 2 layers in LDS.
constexpr int PWL_LROWS = 112;
constexpr int PWL_LTBL = PWL_OFF_PIECE + PWL_LROWS * PWL_ROW;
// The flow kernel (one table set, L <= 3 layers) stages EVERY row instead (L x 46 KB of its 160 KB): no piece is
// ever past the window, and the code that reads rows from the global table is not even compiled in.  The LDS
// window is the template parameter LROWS of everything below.
constexpr int pwl_ltbl(int lrows) { return PWL_OFF_PIECE + lrows * PWL_ROW; }
constexpr int PWL_N_SLOT = PWL_NBP - 1;   // the piece count n, stored (as int bits) in the last padding slot of bp[]


// One block (512 threads) per (slice, layer).  Rows are written in the form the spline that consumes them
// (cond_spline_masked) wants:
//   m = 0..4, 5..9   softmax logits in log2 units, SHIFTED by a per-piece constant per group: the group's largest
//                    logit at the piece's centre is 0 (softmax is shift-invariant).  Where the group maximum
//                    stays within +-PWL_FAST_LOGIT of 0 over the piece, the per-sample running maximum and its
//                    subtraction are not needed: e_k = 2^th_k directly, with the dominant terms' exponents small
//                    (full fp32 accuracy) and no overflow;
//   m = 10..15       slope logits with the softplus offset added, in log2 units.
// The shift-free evaluation is only safe (and only accurate) under those bounds and with the slope logits in
// [PWL_FAST_SLOPE_LO, PWL_FAST_SLOPE_HI] (natural units): the kernel checks this HERE, per piece over the part of
// the piece inside the search grid, and marks every grid cell that touches a piece outside the bounds (and the
// two outermost cells, which also catch every u beyond the grid) with the sign bit of its entry: a wave that
// sees a marked cell evaluates the spline the general way (running maximum, log1p series for tiny slopes).
// Rows past the last piece are never read (the search stops at the +inf padding) and are left unwritten.
// `c_offset` is added to the slice's condition (the loss kernels need t - dt/2 and t + dt/2).
constexpr double PWL_FAST_LOGIT = 4.0;           // log2 units: the group maximum stays in [-4, 4] over the piece
constexpr double PWL_FAST_SLOPE_LO = -3.0;       // below: log(1 + e^v) in fp32 loses the slope's relative accuracy
constexpr double PWL_FAST_SLOPE_HI = 40.0;
constexpr int PWL_CHUNK = 64;             // pieces per pass of the two-stage affine-map computation

#ifndef CNF_PWL_NO_BUILDER          /* (a second translation unit that only READS tables defines this) */
__global__ __launch_bounds__(512) void pwl_build_kernel(const float* __restrict__ weights /* prep + hdr */,
                                                        int64_t per_layer, const float* __restrict__ cvals,
                                                        float c_offset, int L, double sp_offset,
                                                        float* __restrict__ tables) {
  __shared__ double a1[PWL_H], b1v[PWL_H], W1[PWL_H * PWL_H], bb1[PWL_H], Wo[PWL_H * PWL_P], bo[PWL_P];
  __shared__ double bpu[PWL_H], sbp[PWL_H];
  __shared__ double candu[PWL_NPIECE - 1], cand[PWL_NPIECE - 1];
  __shared__ double PQ[PWL_CHUNK * PWL_H * 2];
  __shared__ double RAW[PWL_CHUNK * PWL_P * 2];     // (S, T about u_ref) of every output of the chunk's pieces
  __shared__ int bad[PWL_NPIECE];                  // piece needs the general spline evaluation
  __shared__ double urefs[PWL_CHUNK];
  __shared__ int icnt[17], ioff[17];               // finite roots per first-layer interval, and their prefix sums
  // Second-layer pre-activations P u + Q of every (first-layer interval, unit): inside an interval the first layer's
  // activity pattern is fixed, so every piece of the interval has these coefficients -- stage 1 below looks them up
  // instead of redoing the 16-term sums per (piece, unit).  biv[q]: the interval of the piece to the LEFT of
  // breakpoint q.
  __shared__ double IPQ[17 * PWL_H * 2];
  __shared__ int biv[PWL_NPIECE];
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
  double root = INF;
  const int iv = tid / PWL_H, kk = tid % PWL_H;        // interval, second-layer unit (tid < 17 * 16)
  if (tid < 17 * PWL_H) {
    const double lo = iv == 0 ? -INF : sbp[iv - 1], hi = iv == PWL_H ? INF : sbp[iv];
    if (lo < INF) {             // (an interval between tied breakpoints is empty but still owns a zero-width piece)
      const double u = test_point(lo, hi);
      double P = 0.0, Q = bb1[kk];
      for (int j = 0; j < PWL_H; ++j) {
        const double on = a1[j] * u + b1v[j] > 0.0 ? W1[j * PWL_H + kk] : 0.0;
        P += on * a1[j];
        Q += on * b1v[j];
      }
      IPQ[2 * tid] = P; IPQ[2 * tid + 1] = Q;
      if (lo < hi && P != 0.0) { const double r = -Q / P; if (r > lo && r < hi) root = r; }
    }
    candu[PWL_H + tid] = root;
  }
  for (int p = tid; p < PWL_NPIECE - 1; p += blockDim.x) cand[p] = INF;
  for (int p = tid; p < PWL_NPIECE; p += blockDim.x) bad[p] = 0;
  __syncthreads();
  // Sorted order without a 288 x 288 rank sort: the roots of interval i lie strictly between the sorted
  // first-layer breakpoints sbp[i-1] and sbp[i], so the sorted sequence is, interval by interval, the interval's
  // roots (ranked among the <= 16 of them: 16 comparisons, ties by unit) followed by sbp[i].
  int rank_in = 0;
  if (tid < 17 * PWL_H) {
    int cnt = 0;
    for (int j = 0; j < PWL_H; ++j) {
      const double o = candu[PWL_H + iv * PWL_H + j];
      rank_in += (o < root || (o == root && j < kk)) ? 1 : 0;
      cnt += o < INF ? 1 : 0;
    }
    if (kk == 0) icnt[iv] = cnt;
  }
  __syncthreads();
  if (tid < 17) {                                   // roots before interval tid
    int sum = 0;
    for (int i = 0; i < tid; ++i) sum += icnt[i];
    ioff[tid] = sum;
  }
  __syncthreads();
  if (tid < 17 * PWL_H && root < INF) {          // iv first-layer breakpoints precede it
    cand[iv + ioff[iv] + rank_in] = root;
    biv[iv + ioff[iv] + rank_in] = iv;
  }
  if (tid < PWL_H && sbp[tid] < INF) {
    cand[tid + ioff[tid] + icnt[tid]] = sbp[tid];
    biv[tid + ioff[tid] + icnt[tid]] = tid;       // the piece that ends at the t-th sorted first-layer breakpoint
  }
  int n1 = 0;
  for (int j = 0; j < PWL_H; ++j) n1 += sbp[j] < INF ? 1 : 0;
  const int n = n1 + ioff[16] + icnt[16];             // finite breakpoints; pieces 0 .. n
  __syncthreads();
  float* T = tables + (int64_t)blockIdx.x * PWL_TBL;
  for (int p = tid; p < PWL_NBP; p += blockDim.x)
    // padding: NaN -- `bp <= u` is false for every u, +inf included, so the searches stop there unaided
    T[p] = p == PWL_N_SLOT ? __int_as_float(n) : (p < n ? (float)cand[p] : __int_as_float(0x7fc00000));
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
      const int ivp = p < n ? biv[p] : n1;          // the last piece lies beyond every finite first-layer breakpoint
      const double P = IPQ[2 * (ivp * PWL_H + k)], Q = IPQ[2 * (ivp * PWL_H + k) + 1];
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
      if (m == 0) {        // also in the row's first padding slot: it then arrives with the row's own LDS reads
        T[PWL_OFF_REF + p] = uref; urefs[pl] = (double)uref;
        T[PWL_OFF_PIECE + p * PWL_ROW + 2 * PWL_P] = uref;
      }
      RAW[2 * t] = S;
      RAW[2 * t + 1] = Tt;
    }
    __syncthreads();
    for (int t = tid; t < np * PWL_P; t += blockDim.x) {
      const int pl = t >> 4, m = t & 15, p = base + pl;
      double S = RAW[2 * t], Tt = RAW[2 * t + 1];
      const double lo = p == 0 ? -INF : cand[p - 1], hi = p < n ? cand[p] : INF;
      const double uref = urefs[pl];
      // the part of the piece a sample can reach through an unmarked grid cell
      const double ulo = fmax(lo, (double)PWL_GMIN - 0.01), uhi = fmin(hi, -(double)PWL_GMIN + 0.01);
      bool out_of_bounds = false;
      if (m < 2 * 5) {
        // shift of the group: its largest logit at the centre of the reachable part of the piece; bounds: the
        // group maximum at both ends of that part (linear logits: the extremes of the maximum are at the ends
        // or at the centre, where it is 0 by construction -- the maximum of linear functions is convex)
        const int r0 = m < 5 ? 0 : 5;
        const double uc = ulo <= uhi ? 0.5 * (ulo + uhi) : uref;
        double Mc = -INF, Mlo = -INF, Mhi = -INF;
        for (int k = r0; k < r0 + 5; ++k) {
          const double Sk = RAW[2 * (pl * PWL_P + k)], Tk = RAW[2 * (pl * PWL_P + k) + 1];
          Mc = fmax(Mc, Tk + Sk * (uc - uref));
          Mlo = fmax(Mlo, Tk + Sk * (ulo - uref));
          Mhi = fmax(Mhi, Tk + Sk * (uhi - uref));
        }
        if (ulo <= uhi)
          out_of_bounds = fmax(fabs(Mlo - Mc), fabs(Mhi - Mc)) * LOG2E_D > PWL_FAST_LOGIT;
        S *= LOG2E_D;
        Tt = (Tt - Mc) * LOG2E_D;
      } else {
        Tt += sp_offset;
        if (ulo <= uhi) {
          const double a = Tt + S * (ulo - uref), b = Tt + S * (uhi - uref);
          out_of_bounds = fmin(a, b) < PWL_FAST_SLOPE_LO || fmax(a, b) > PWL_FAST_SLOPE_HI;
        }
        S *= LOG2E_D; Tt *= LOG2E_D;
      }
      if (out_of_bounds || !(S == S) || !(Tt == Tt)) bad[p] = 1;      // (benign race: everybody writes 1)
      float* row = T + PWL_OFF_PIECE + p * PWL_ROW;
      row[m] = (float)S;
      row[PWL_P + m] = (float)Tt;
    }
    __syncthreads();
  }
  // search grid: number of breakpoints <= the cell's left edge (a lower bound for the search).  The edge is
  // pulled in by 1e-4: pwl_cell() computes the cell in fp32, and u a rounding error below an edge may land
  // in the cell above it.  PWL_G_MANY: more than two breakpoints in (left edge, right edge] -- the kernels compare
  // against two breakpoints unconditionally and loop only in such cells (cell 0 / the last cell also serve every
  // u beyond the grid).  PWL_G_GENERAL: the cell touches a piece that needs the general spline evaluation, or is
  // one of the two outermost cells.
  // lo_ of cell g = the number of breakpoints b <= x_g = GMIN + g / GSCALE - 1e-4 -- counted from the breakpoints'
  // side: breakpoint b is below the left edge of every cell from g_b = ceil((b + 1e-4 - GMIN) GSCALE) on, so each
  // breakpoint adds one at g_b and lo_ is the running sum (a block scan: four consecutive cells per thread, a
  // shuffle scan inside the wave, the eight wave totals through LDS).  Round 2's first version searched every cell
  // through nine dependent LDS reads: the largest phase of the kernel.  A cell index off by one through rounding is
  // covered by the 1e-4 the edge is pulled in by (see above).
  int* delta = reinterpret_cast<int*>(PQ);               // PQ is free again: 2 049 + 8 ints
  static_assert(sizeof(double) * PWL_CHUNK * PWL_H * 2 >= sizeof(int) * (PWL_NG + 16), "scratch for the grid scan");
  constexpr int CPT = 4;
  static_assert(PWL_NG == 512 * CPT, "one thread owns four consecutive cells (512 threads)");
  for (int g = tid; g < PWL_NG + 16; g += blockDim.x) delta[g] = 0;
  __syncthreads();
  for (int p = tid; p < n; p += blockDim.x) {
    const double gb = ceil((cand[p] + 1e-4 - (double)PWL_GMIN) * (double)PWL_GSCALE);
    // (cell 0 serves every u below the grid and searches from piece 0: breakpoints below the grid count from cell 1)
    const int g0 = gb < 1.0 ? 1 : (gb >= (double)PWL_NG ? PWL_NG : (int)gb);
    atomicAdd(&delta[g0], 1);                             // (slot PWL_NG: breakpoints beyond the grid, never summed)
  }
  __syncthreads();
  {
    const int g0 = tid * CPT;
    int c0 = delta[g0], c1 = c0 + delta[g0 + 1], c2 = c1 + delta[g0 + 2], c3 = c2 + delta[g0 + 3];
    int run = c3;                                         // inclusive scan of the threads' totals over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(run, off, 64);
      if ((tid & 63) >= off) run += v;
    }
    int* wtot = delta + PWL_NG + 8;                       // eight wave totals
    __syncthreads();                                      // (everybody has read its delta entries)
    if ((tid & 63) == 63) wtot[tid >> 6] = run;
    __syncthreads();
    int before = run - c3;                                // exclusive within the wave
    for (int w2 = 0; w2 < (tid >> 6); ++w2) before += wtot[w2];
    const int cnt[CPT] = {before + c0, before + c1, before + c2, before + c3};
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int g = g0 + i;
      const int lo_ = g == 0 ? 0 : cnt[i];
      const double xr = g == PWL_NG - 1 ? INF : (double)PWL_GMIN + (double)(g + 1) / (double)PWL_GSCALE + 1e-4;
      int mark = (g == 0 || g == PWL_NG - 1) ? 1 : 0;
      for (int p = lo_; p <= n && !mark; ++p) {          // pieces lo_ .. the one holding the cell's right edge
        mark |= bad[p];
        if (p < n && cand[p] > xr) break;
      }
      const bool many = lo_ + 2 < n && cand[lo_ + 2] <= xr;      // a third breakpoint a sample of the cell can pass
      reinterpret_cast<uint16_t*>(T + PWL_OFF_GRID)[g] =
          (uint16_t)((uint32_t)lo_ | (mark ? PWL_G_GENERAL : 0u) | (many ? PWL_G_MANY : 0u));
    }
  }
}

#endif  // CNF_PWL_NO_BUILDER

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
template <int LROWS = PWL_LROWS>
__device__ __forceinline__ void pwl_row(const float* tbl, const float* __restrict__ gtbl, int p, float u,
                                        float (&th)[PWL_P]) {
  const float du = u - tbl[PWL_OFF_REF + p];
  if (LROWS >= PWL_NPIECE || p < LROWS) {
    // ONE address per sample; the 8 chunks are immediate offsets of the ds_read_b128s
    const lds_f4_ptr row = (lds_f4_ptr)(uintptr_t)((uint32_t)(uintptr_t)(tbl + PWL_OFF_PIECE) + __umul24((uint32_t)p, PWL_ROW * 4));
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

// `general`: the sample's grid cell is marked -- the spline must be evaluated the general way (pwl_build_kernel)
// Piece of u: grid lookup, two unrolled comparisons, and a loop only where the builder found more than two
// breakpoints in reach of the cell.
__device__ __forceinline__ int pwl_piece(const float* tbl, float u, bool& general) {
  const uint32_t g = reinterpret_cast<const uint16_t*>(tbl + PWL_OFF_GRID)[pwl_cell(u)];
  general = (g & PWL_G_GENERAL) != 0;
  int p = (int)(g & PWL_G_INDEX);
  const float us = u;                            // (the padding is NaN: no comparison with it succeeds)
  const float b0 = tbl[p], b1 = tbl[p + 1];      // sorted: b1 <= us implies b0 <= us
  p += (b0 <= us ? 1 : 0) + (b1 <= us ? 1 : 0);
  if (g & PWL_G_MANY) while (tbl[p] <= us) ++p;  // bp[>= n] = NaN: terminates
  return p;
}
// Two samples: one wave-level branch for the rare loop.
__device__ __forceinline__ void pwl_piece(const float* tbl, v2f u, int& px, int& py, bool& general) {
  const uint16_t* grid = reinterpret_cast<const uint16_t*>(tbl + PWL_OFF_GRID);
  const uint32_t gx = grid[pwl_cell(u.x)], gy = grid[pwl_cell(u.y)];
  general = ((gx | gy) & PWL_G_GENERAL) != 0;
  px = (int)(gx & PWL_G_INDEX); py = (int)(gy & PWL_G_INDEX);
  const float ux = u.x, uy = u.y;
  const float ax0 = tbl[px], ax1 = tbl[px + 1], ay0 = tbl[py], ay1 = tbl[py + 1];
  px += (ax0 <= ux ? 1 : 0) + (ax1 <= ux ? 1 : 0);
  py += (ay0 <= uy ? 1 : 0) + (ay1 <= uy ? 1 : 0);
  if ((gx | gy) & PWL_G_MANY) {
    bool more;
    do {
      const bool mx = tbl[px] <= ux, my = tbl[py] <= uy;
      px += mx ? 1 : 0;
      py += my ? 1 : 0;
      more = mx || my;
    } while (more);
  }
}

// `general`: the sample's grid cell is marked -- the spline must be evaluated the general way (pwl_build_kernel)
template <int LROWS = PWL_LROWS>
__device__ __forceinline__ void pwl_eval(const float* tbl, const float* __restrict__ gtbl, float u, float (&th)[PWL_P],
                                         bool& general) {
  const int p = pwl_piece(tbl, u, general);
  pwl_row<LROWS>(tbl, gtbl, p, u, th);
}

template <int LROWS = PWL_LROWS>
__device__ __forceinline__ void pwl_eval(const float* tbl, const float* __restrict__ gtbl, v2f u, v2f (&th)[PWL_P],
                                         bool& general) {
  int px, py;
  pwl_piece(tbl, u, px, py, general);
  float tx[PWL_P], ty[PWL_P];
  pwl_row<LROWS>(tbl, gtbl, px, u.x, tx);
  pwl_row<LROWS>(tbl, gtbl, py, u.y, ty);
#pragma unroll
  for (int m = 0; m < PWL_P; ++m) th[m] = v2f{tx[m], ty[m]};
}

// --- rows evaluated in two steps (cond_spline_rows): the softmax logits as parameter pairs, then -- once the bin is
// known -- the bin's two slope logits.  Same rows, same FMAs as pwl_row.  The LDS reads are unconditional (from a
// clamped row for pieces past the LDS window, whose values are then replaced from the global table in a branch
// that is almost never taken): the scheduler is free to issue them early and no lane waits inside a branch.
typedef const float __attribute__((address_space(3))) * lds_f_ptr;
typedef const v2f __attribute__((address_space(3))) * lds_v2_ptr;

struct PwlRows {
  int pa, pb;          // piece of sample a / b
  float dua, dub;      // u - u_ref of the piece (set by pwl_logit_pairs: u_ref is read with the row)
  lds_f_ptr ra, rb;    // the piece's row in LDS (row LROWS - 1 for pieces past the window)
};

template <int LROWS>
__device__ __forceinline__ lds_f_ptr pwl_lds_row(const float* tbl, int p) {
  const uint32_t q = LROWS >= PWL_NPIECE ? (uint32_t)p : (uint32_t)(p < LROWS ? p : LROWS - 1);
  static_assert(PWL_ROW == 36, "row stride 144 B = 9 << 4");
  uint32_t q9 = (q << 3) + q;                      // two full-rate shifts-and-adds instead of v_mul_lo_u32
  asm volatile("" : "+v"(q9));
  return (lds_f_ptr)(uintptr_t)((uint32_t)(uintptr_t)(tbl + PWL_OFF_PIECE) + (q9 << 4));
}

template <int LROWS>
__device__ __forceinline__ void pwl_find(const float* tbl, v2f u, PwlRows& r, bool& general) {
  int px, py;
  pwl_piece(tbl, u, px, py, general);
  r.pa = px; r.pb = py;
  r.ra = pwl_lds_row<LROWS>(tbl, px);
  r.rb = pwl_lds_row<LROWS>(tbl, py);
}

template <class FP, class F4P, class V2P>
__device__ __forceinline__ float pwl_logit_pairs_(FP r, F4P r4, V2P r2, float u, v2f (&q)[5]) {
  const f4 s0 = r4[0], s1 = r4[1], t0 = r4[4], t1 = r4[5];
  const v2f s2 = r2[4], t2 = r2[12];
  const float du = u - r[2 * PWL_P];
  const v2f d2 = v2f{du, du};
  q[0] = __builtin_elementwise_fma(__builtin_shufflevector(s0, s0, 0, 1), d2, __builtin_shufflevector(t0, t0, 0, 1));
  q[1] = __builtin_elementwise_fma(__builtin_shufflevector(s0, s0, 2, 3), d2, __builtin_shufflevector(t0, t0, 2, 3));
  q[2] = __builtin_elementwise_fma(__builtin_shufflevector(s1, s1, 0, 1), d2, __builtin_shufflevector(t1, t1, 0, 1));
  q[3] = __builtin_elementwise_fma(__builtin_shufflevector(s1, s1, 2, 3), d2, __builtin_shufflevector(t1, t1, 2, 3));
  q[4] = __builtin_elementwise_fma(s2, d2, t2);
  return du;
}

// the 10 softmax logits of piece p at u, as pairs (2j, 2j+1); returns u - u_ref
template <int LROWS>
__device__ __forceinline__ float pwl_logit_pairs(lds_f_ptr row, const float* __restrict__ gtbl, int p, float u,
                                                 v2f (&q)[5]) {
  float du = pwl_logit_pairs_(row, (lds_f4_ptr)row, (lds_v2_ptr)row, u, q);
  if (LROWS < PWL_NPIECE && p >= LROWS) {
    const float* g = gtbl + PWL_OFF_PIECE + p * PWL_ROW;
    du = pwl_logit_pairs_(g, reinterpret_cast<const f4*>(g), reinterpret_cast<const v2f*>(g), u, q);
  }
  return du;
}

// (t_k, t_k+1): the slope logits of bin k's two knots (k in 0 .. 4)
template <int LROWS>
__device__ __forceinline__ v2f pwl_slope_pair(lds_f_ptr row, const float* __restrict__ gtbl, int p, int k, float du) {
  const v2f d2 = v2f{du, du};
  const lds_f_ptr r = row + k;
  v2f t = __builtin_elementwise_fma(v2f{r[10], r[11]}, d2, v2f{r[PWL_P + 10], r[PWL_P + 11]});
  if (LROWS < PWL_NPIECE && p >= LROWS) {
    const float* g = gtbl + PWL_OFF_PIECE + p * PWL_ROW + k;
    t = __builtin_elementwise_fma(v2f{g[10], g[11]}, d2, v2f{g[PWL_P + 10], g[PWL_P + 11]});
  }
  return t;
}

// Stage the L tables of (set, slice) into LDS: header arrays + the rows in use, at most LROWS.
template <int LROWS = PWL_LROWS>
__device__ __forceinline__ void pwl_stage(float* tbl, const float* __restrict__ g0, int L, int tid, int nthreads) {
  for (int l = 0; l < L; ++l) {
    const float* g = g0 + (int64_t)l * PWL_TBL;
    const int n = __float_as_int(g[PWL_N_SLOT]);                       // pieces 0 .. n
    const int rows = n + 1 < LROWS ? n + 1 : LROWS;
    const f4* src = reinterpret_cast<const f4*>(g);
    f4* dst = reinterpret_cast<f4*>(tbl + l * pwl_ltbl(LROWS));
    for (int i = tid; i < (PWL_OFF_PIECE + rows * PWL_ROW) / 4; i += nthreads) dst[i] = src[i];
  }
}

}  // namespace cnf
