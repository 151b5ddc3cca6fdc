/*
 * cnf_oracle.h -- C interface of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * See cnf_oracle_impl.h for what is restated, from which reference lines, and
 * for the parity status ("parity unpinned" in absolute value vs distrax).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 */
#ifndef CNF_ORACLE_H
#define CNF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNF_ORACLE_MAX_D 64
#define CNF_ORACLE_MAX_H 256
#define CNF_ORACLE_MAX_K 64

typedef struct {
  int32_t D;  /* event dim            (config general.dim)            */
  int32_t L;  /* flow layers          (cnf.flow_num_layers)           */
  int32_t H;  /* hidden width         (cnf.hidden_size)               */
  int32_t M;  /* hidden layers        (cnf.mlp_num_layers), >= 1      */
  int32_t K;  /* spline bins          (cnf.num_bins)                  */
  double range_min, range_max;   /* flows.py:127-128: -10, 10         */
  double min_bin_size;           /* distrax default 1e-4              */
  double min_knot_slope;         /* flows.py:130: 1e-4                */
  int32_t periodized;            /* flows.py:58-64,127-131: sin/cos features of the conditioner input,
                                    range [0, 2 pi], boundary_slopes='circular' (last knot slope := first) */
} cnf_oracle_cfg;

int cnf_oracle_check_cfg(const cnf_oracle_cfg *g);
size_t cnf_oracle_param_count(const cnf_oracle_cfg *g);

/* c value of sample i is c[i / c_block]. */
#define CNF_ORACLE_DECL(REAL, SFX)                                              \
  int cnf_oracle_forward_logdet##SFX(const cnf_oracle_cfg *, const REAL *params,\
      const REAL *x, const REAL *c, int64_t c_block, REAL *y, REAL *logdet,     \
      int64_t B);                                                               \
  int cnf_oracle_inverse_logdet##SFX(const cnf_oracle_cfg *, const REAL *params,\
      const REAL *y, const REAL *c, int64_t c_block, REAL *x, REAL *logdet,     \
      int64_t B);                                                               \
  int cnf_oracle_log_prob##SFX(const cnf_oracle_cfg *, const REAL *params,      \
      const REAL *value, const REAL *c, int64_t c_block, REAL *logp, int64_t B);\
  int cnf_oracle_sample_logprob##SFX(const cnf_oracle_cfg *, const REAL *params,\
      const REAL *noise, const REAL *c, int64_t c_block, REAL *y, REAL *logp,   \
      int64_t B);                                                               \
  int cnf_oracle_rqs##SFX(const REAL *theta, const REAL *v, int64_t n, int K,   \
      REAL lo, REAL hi, REAL min_bin, REAL min_slope, int inverse, REAL *out,   \
      REAL *logdet);                                                            \
  int cnf_oracle_knots##SFX(const REAL *theta, int K, REAL lo, REAL hi,         \
      REAL min_bin, REAL min_slope, REAL *xk, REAL *yk, REAL *dl);

CNF_ORACLE_DECL(double, _f64)
CNF_ORACLE_DECL(float, _f32)

/* Philox4x32-10 standard-normal stream shared with the HIP side: flat element
 * e = (offset + i) * D + d of the [*, D] noise tensor comes from counter block
 * e >> 2 under key (seed_lo, seed_hi); see cnf_oracle.c. */
void cnf_oracle_philox4x32(const uint32_t ctr[4], const uint32_t key[2],
                           uint32_t out[4]);
int cnf_oracle_normal_f64(uint64_t seed, uint64_t first_element, int64_t n,
                          double *out);

int cnf_oracle_num_threads(void);
void cnf_oracle_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
/* Threefry-2x32-20 and the JAX-style float64 normal draw built on it (see cnf_oracle.c) */
void cnf_oracle_threefry2x32(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2]);
int cnf_oracle_normal_threefry_f64(uint32_t key0, uint32_t key1, uint64_t size, uint64_t first_element, int64_t n,
                                   double *out);

#endif
