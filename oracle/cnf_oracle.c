/*
 * cnf_oracle.c -- CPU oracle for the conditional RQS flow path
 * (TEST INFRASTRUCTURE ONLY; see cnf_oracle_impl.h for citations and the
 * "parity unpinned" statement).  Build: make -C oracle
 */
#include "cnf_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int cnf_oracle_check_cfg(const cnf_oracle_cfg *g) {
  if (!g) return -1;
  if (g->D < 1 || g->D > CNF_ORACLE_MAX_D) return -1;
  if (g->L < 1) return -1;
  if (g->H < 1 || g->H > CNF_ORACLE_MAX_H) return -1;
  if (g->M < 1) return -1;
  if (g->K < 1 || g->K > CNF_ORACLE_MAX_K) return -1;
  if (!(g->range_min < g->range_max)) return -1;
  if (!(g->min_bin_size > 0) || !(g->min_knot_slope > 0) || !(g->min_knot_slope < 1)) return -1;
  if (g->K * g->min_bin_size > g->range_max - g->range_min) return -1;
  return 0;
}

size_t cnf_oracle_param_count(const cnf_oracle_cfg *g) {
  size_t H = (size_t)g->H, P = (size_t)(3 * g->K + 1), n = P;
  for (int d = 1; d < g->D; ++d)
    n += (size_t)g->L * ((size_t)(g->periodized ? 2 : 1) * (size_t)(1 + d) * H + H +
                         (size_t)(g->M - 1) * (H * H + H) + H * P + P);
  return n;
}

void cnf_oracle_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int cnf_oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---- float64 instance (reference dtype, solvers.py:23) ---- */
#define REAL double
#define FN(name) name##_f64
#define R_EXP exp
#define R_LOG log
#define R_LOG1P log1p
#define R_SQRT sqrt
#define R_FABS fabs
#define R_SIN sin
#define R_COS cos
#include "cnf_oracle_impl.h"
#undef R_SIN
#undef R_COS
#undef REAL
#undef FN
#undef R_EXP
#undef R_LOG
#undef R_LOG1P
#undef R_SQRT
#undef R_FABS

/* ---- float32 instance (CPU baseline in the kernels' dtype) ---- */
#define REAL float
#define FN(name) name##_f32
#define R_EXP expf
#define R_LOG logf
#define R_LOG1P log1pf
#define R_SQRT sqrtf
#define R_FABS fabsf
#define R_SIN sinf
#define R_COS cosf
#include "cnf_oracle_impl.h"
#undef REAL
#undef FN
#undef R_SIN
#undef R_COS
#undef R_EXP
#undef R_LOG
#undef R_LOG1P
#undef R_SQRT
#undef R_FABS

/* ---- Philox4x32-10 (Salmon et al. 2011) + Box-Muller ----------------------
 * The reference draws base noise with jax.random (conditional.py:378,399), a
 * JAX-version-dependent threefry stream that cannot be reproduced here; the
 * build defines its own counter-based stream so that noise is a pure function
 * of (seed, global element index), independent of the number of GPUs. */
void cnf_oracle_philox4x32(const uint32_t ctr[4], const uint32_t key[2],
                           uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

int cnf_oracle_normal_f64(uint64_t seed, uint64_t first_element, int64_t n,
                          double *out) {
  const double two_pi = 6.283185307179586476925286766559;
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  for (int64_t i = 0; i < n; ++i) {
    uint64_t e = first_element + (uint64_t)i, blk = e >> 2;
    uint32_t ctr[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), 0u, 0u}, u[4];
    cnf_oracle_philox4x32(ctr, key, u);
    int r = (int)(e & 3), p = r >> 1;
    double u1 = (double)((u[2 * p] >> 8) + 1u) * (1.0 / 16777216.0);
    double u2 = (double)(u[2 * p + 1] >> 8) * (1.0 / 16777216.0);
    double rad = sqrt(-2.0 * log(u1)), ang = two_pi * u2;
    out[i] = (r & 1) ? rad * sin(ang) : rad * cos(ang);
  }
  return 0;
}

/* ---- Threefry-2x32-20 (Salmon et al. 2011, Random123) and the JAX-style normal draw --------------------------
 * Restates jax._src.prng.threefry_2x32 / threefry_random_bits (classic, non-partitionable path) and
 * jax._src.random._uniform / _normal_real for float64: see cnf_ot_amd/csrc/cnf_flow.hip fill_normal_threefry_kernel.
 * The block function is pinned by the Random123 known-answer vectors (tests/test_oracle_flow.py). */
static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

void cnf_oracle_threefry2x32(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2]) {
  const uint32_t ks[3] = {key[0], key[1], key[0] ^ key[1] ^ 0x1BD11BDAu};
  static const int rot[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  uint32_t x0 = ctr[0] + ks[0], x1 = ctr[1] + ks[1];
  for (int g = 0; g < 5; ++g) {
    for (int r = 0; r < 4; ++r) { x0 += x1; x1 = rotl32(x1, rot[g & 1][r]); x1 ^= x0; }
    x0 += ks[(g + 1) % 3];
    x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
  out[0] = x0; out[1] = x1;
}

/* erfinv by Newton-Halley on libm's erf from a rational start (glibc has no erfinv) */
static double erfinv_newton(double y) {
  if (y <= -1.0) return -INFINITY;
  if (y >= 1.0) return INFINITY;
  double w = -log((1.0 - y) * (1.0 + y)), x;
  if (w < 5.0) {
    w -= 2.5;
    x = 2.81022636e-08; x = 3.43273939e-07 + x * w; x = -3.5233877e-06 + x * w; x = -4.39150654e-06 + x * w;
    x = 0.00021858087 + x * w; x = -0.00125372503 + x * w; x = -0.00417768164 + x * w; x = 0.246640727 + x * w;
    x = 1.50140941 + x * w;
  } else {
    w = sqrt(w) - 3.0;
    x = -0.000200214257; x = 0.000100950558 + x * w; x = 0.00134934322 + x * w; x = -0.00367342844 + x * w;
    x = 0.00573950773 + x * w; x = -0.0076224613 + x * w; x = 0.00943887047 + x * w; x = 1.00167406 + x * w;
    x = 2.83297682 + x * w;
  }
  x *= y;
  for (int it = 0; it < 3; ++it) {
    const double e = erf(x) - y, d = 1.1283791670955126 * exp(-x * x);      /* 2/sqrt(pi) exp(-x^2) */
    x -= e / (d + x * e);                                                    /* Halley: f'' / f' = -2x */
  }
  return x;
}

int cnf_oracle_normal_threefry_f64(uint32_t key0, uint32_t key1, uint64_t size, uint64_t first_element, int64_t n,
                                   double *out) {
  const double lo = nextafter(-1.0, 0.0);
  const uint32_t key[2] = {key0, key1};
  if (first_element + (uint64_t)n > size || size > 0x7fffffffull) return -1;
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t j = first_element + (uint64_t)i;
    const uint32_t ctr[2] = {(uint32_t)j, (uint32_t)(size + j)};
    uint32_t o[2];
    cnf_oracle_threefry2x32(key, ctr, o);
    const uint64_t bits = (((uint64_t)o[0] << 32) | (uint64_t)o[1]) >> 12 | 0x3FF0000000000000ull;
    double f;
    memcpy(&f, &bits, sizeof f);
    f -= 1.0;
    double u = f * (1.0 - lo) + lo;
    if (u < lo) u = lo;
    out[i] = 1.41421356237309504880 * erfinv_newton(u);
  }
  return 0;
}
