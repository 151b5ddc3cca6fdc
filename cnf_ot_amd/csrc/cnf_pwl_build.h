// cnf_pwl_build.h -- the BUILDER of the dim-2 conditioner tables (pwl_build_kernel and its helpers): included by the
// one translation unit that launches it (cnf_flow.hip).  The table layout and the readers are in cnf_pwl.h.
#pragma once

#include "cnf_pwl.h"

namespace cnf {

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


}  // namespace cnf
