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

// LN2 x (d theta / d u of piece p) . theta_bar for one sample: the 2K softmax entries `t` and the two slope entries
// (s0, s1) at knots k, k + 1 -- the conditioner's input adjoint of the table backward (rows are in log2 units)
template <int LROWS, int NT>
__device__ __forceinline__ float pwl_row_dot(lds_f_ptr row, const float* __restrict__ gtbl, int p, int k,
                                             const float (&t)[NT], float s0, float s1) {
  static_assert(NT == 10, "K = 5");
  auto dot = [&](f4 a, f4 b, v2f c, float k0, float k1) {
    float r = k0 * s0;
    r = fmaf(k1, s1, r);
#pragma unroll
    for (int e = 0; e < 4; ++e) { r = fmaf(a[e], t[e], r); r = fmaf(b[e], t[4 + e], r); }
    r = fmaf(c.x, t[8], r);
    return fmaf(c.y, t[9], r) * 0.693147182464599609375f;
  };
  const lds_f_ptr rk = row + k;
  float r = dot(((lds_f4_ptr)row)[0], ((lds_f4_ptr)row)[1], ((lds_v2_ptr)row)[4], rk[10], rk[11]);
  if (LROWS < PWL_NPIECE && p >= LROWS) {
    const float* g = gtbl + PWL_OFF_PIECE + p * PWL_ROW;
    r = dot(reinterpret_cast<const f4*>(g)[0], reinterpret_cast<const f4*>(g)[1], reinterpret_cast<const v2f*>(g)[4], g[10 + k], g[11 + k]);
  }
  return r;
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
