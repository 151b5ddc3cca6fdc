/*
 * cnf_oracle.c -- CPU oracle for the conditional RQS flow path
 * (TEST INFRASTRUCTURE ONLY; see cnf_oracle_impl.h for citations and the
 * "parity unpinned" statement).  Build: make -C oracle
 */
#include "cnf_oracle.h"

#include <math.h>
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
    n += (size_t)g->L * ((size_t)(1 + d) * H + H + (size_t)(g->M - 1) * (H * H + H) + H * P + P);
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
#include "cnf_oracle_impl.h"
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
#include "cnf_oracle_impl.h"
#undef REAL
#undef FN
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
