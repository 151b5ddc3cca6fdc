/*
 * cnf_oracle_impl.h -- CPU restatement of cnf_ot's conditional RQS flow path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: it
 * is the checker for the HIP path (tests/, __graft_entry__.smoke()) and the
 * "port" CPU baseline of bench.py.  The product path never calls it.
 *
 * PARITY STATUS: "parity unpinned" in absolute value against distrax.  The
 * spline arithmetic of the reference lives in the un-vendored, un-pinned
 * third-party package distrax (distrax.RationalQuadraticSpline, call site
 * /root/reference/cnf_ot/models/flows.py:124-132); jax/distrax/haiku are not
 * installed in the build container, and the reference holds no golden vectors
 * for this path (tests/test_rqs_accuracy.py:18-172 is property-only).  This
 * file restates the published algorithm (Durkan et al. 2019, as implemented by
 * distrax/_src/bijectors/rational_quadratic_spline.py)
 * and is pinned by (i) the property suite of the reference's own test,
 * (ii) identity-at-init (flows.py:48,71-76), (iii) the forward formula that
 * the reference's cnf_ot/models/nsf_symbol.py:6-13 differentiates (that file
 * runs here; its output is a committed fixture).
 *
 * This header is included twice by cnf_oracle.c, once with REAL=double (the
 * reference dtype, solvers.py:23) and once with REAL=float.
 *
 * Required macros: REAL, FN(name) (symbol suffixing), R_EXP, R_LOG, R_LOG1P,
 * R_SQRT, R_FABS.
 */

/* ---- flat parameter layout (haiku tree order) -------------
 *   first[P]                                     '~'/first, shape (1,P)
 *   for l in 0..L-1, for d in 1..D-1:            name = layer{l}_d{d}
 *     W0[(1+d)*H] row-major [in][out], b0[H]     mlp_{name}/~/linear_0 {w,b}
 *     (M-1) x { Wm[H*H], bm[H] }                 mlp_{name}/~/linear_{m}
 *     Wo[H*P], bo[P]                             linear_out_{name} {w,b}
 * P = 3K+1 (flows.py:134).
 */

static size_t FN(cond_size)(const cnf_oracle_cfg *g, int d) {
  size_t H = (size_t)g->H, P = (size_t)(3 * g->K + 1);
  /* periodized: the MLP sees [sin(c, v), cos(c, v)] (flows.py:58-64, num_fourier_feat = 1): 2 (1 + d) inputs */
  return (size_t)(g->periodized ? 2 : 1) * (size_t)(1 + d) * H + H + (size_t)(g->M - 1) * (H * H + H) + H * P + P;
}

static size_t FN(cond_offset)(const cnf_oracle_cfg *g, int l, int d) {
  size_t off = (size_t)(3 * g->K + 1);
  size_t per_layer = 0;
  for (int dd = 1; dd < g->D; ++dd) per_layer += FN(cond_size)(g, dd);
  off += (size_t)l * per_layer;
  for (int dd = 1; dd < d; ++dd) off += FN(cond_size)(g, dd);
  return off;
}

/* distrax _normalize_bin_sizes / _normalize_knot_slopes / __init__ knot build
 * (constants fixed by flows.py:124-132). */
static void FN(rqs_knots_bs)(const REAL *theta, int K, REAL lo, REAL hi,
                             REAL min_bin, REAL min_slope, int circular, REAL *xk, REAL *yk,
                             REAL *dl);
static void FN(rqs_knots)(const REAL *theta, int K, REAL lo, REAL hi,
                          REAL min_bin, REAL min_slope, REAL *xk, REAL *yk,
                          REAL *dl) {
  FN(rqs_knots_bs)(theta, K, lo, hi, min_bin, min_slope, 0, xk, yk, dl);
}
/* circular: distrax boundary_slopes='circular' (flows.py:131): the unnormalized slope of the last knot is
 * replaced by that of the first before normalisation, so f'(range_min) = f'(range_max). */
static void FN(rqs_knots_bs)(const REAL *theta, int K, REAL lo, REAL hi,
                             REAL min_bin, REAL min_slope, int circular, REAL *xk, REAL *yk,
                             REAL *dl) {
  REAL total = (hi - lo) - (REAL)K * min_bin;
  for (int part = 0; part < 2; ++part) {
    const REAL *u = theta + part * K;
    REAL *pos = part == 0 ? xk : yk;
    REAL mx = u[0];
    for (int k = 1; k < K; ++k) mx = u[k] > mx ? u[k] : mx;
    REAL sum = 0;
    for (int k = 0; k < K; ++k) sum += R_EXP(u[k] - mx);
    REAL run = 0;
    pos[0] = lo;
    for (int k = 0; k < K - 1; ++k) {
      REAL w = R_EXP(u[k] - mx) / sum * total + min_bin;
      run += w;                 /* jnp.cumsum(bin_sizes[..., :-1]) */
      pos[k + 1] = lo + run;
    }
    pos[K] = hi;                /* padded with range_max exactly */
  }
  /* offset = log(exp(1 - m) - 1); softplus(u + offset) + m */
  REAL offset = R_LOG(R_EXP((REAL)1 - min_slope) - (REAL)1);
  for (int k = 0; k <= K; ++k) {
    REAL v = theta[2 * K + (circular && k == K ? 0 : k)] + offset;
    REAL sp = (v > 0 ? v : (REAL)0) + R_LOG1P(R_EXP(-R_FABS(v)));
    dl[k] = sp + min_slope;
  }
}

/* bin selection by mask (distrax: correct_bin = (v>=pos[:-1]) & (v<pos[1:]);
 * no match -> first bin; the tails overwrite the result afterwards). */
static int FN(rqs_bin)(REAL v, const REAL *pos, int K) {
  for (int k = 0; k < K; ++k)
    if (v >= pos[k] && v < pos[k + 1]) return k;
  return 0;
}

static REAL FN(clip01)(REAL z) { return z < 0 ? (REAL)0 : (z > 1 ? (REAL)1 : z); }

/* distrax _rational_quadratic_spline_fwd (reference call: autoregressive.py:100) */
static void FN(rqs_fwd)(REAL x, const REAL *xk, const REAL *yk, const REAL *dl,
                        int K, REAL *y_out, REAL *ld_out) {
  int k = FN(rqs_bin)(x, xk, K);
  REAL x0 = xk[k], x1 = xk[k + 1], y0 = yk[k], y1 = yk[k + 1];
  REAL d0 = dl[k], d1 = dl[k + 1];
  REAL bw = x1 - x0, bh = y1 - y0, s = bh / bw;
  REAL z = FN(clip01)((x - x0) / bw);
  REAL sq_z = z * z, z1mz = z - sq_z, sq_1mz = ((REAL)1 - z) * ((REAL)1 - z);
  REAL st = d1 + d0 - (REAL)2 * s;
  REAL num = bh * (s * sq_z + d0 * z1mz);
  REAL den = s + st * z1mz;
  REAL y = y0 + num / den;
  REAL ld = (REAL)2 * R_LOG(s) +
            R_LOG(d1 * sq_z + (REAL)2 * s * z1mz + d0 * sq_1mz) -
            (REAL)2 * R_LOG(den);
  if (x <= xk[0]) { y = (x - xk[0]) * dl[0] + yk[0]; ld = R_LOG(dl[0]); }
  if (x >= xk[K]) { y = (x - xk[K]) * dl[K] + yk[K]; ld = R_LOG(dl[K]); }
  *y_out = y;
  *ld_out = ld;
}

/* distrax _rational_quadratic_spline_inv (reference call: autoregressive.py:130) */
static void FN(rqs_inv)(REAL y, const REAL *xk, const REAL *yk, const REAL *dl,
                        int K, REAL *x_out, REAL *ld_out) {
  int k = FN(rqs_bin)(y, yk, K);
  REAL x0 = xk[k], x1 = xk[k + 1], y0 = yk[k], y1 = yk[k + 1];
  REAL d0 = dl[k], d1 = dl[k + 1];
  REAL bw = x1 - x0, bh = y1 - y0, s = bh / bw;
  REAL w = FN(clip01)((y - y0) / bh);
  REAL st = d1 + d0 - (REAL)2 * s;
  REAL c = -s * w;
  REAL b = d0 - st * w;
  REAL a = s - b;
  REAL z = -(REAL)2 * c / (b + R_SQRT(b * b - (REAL)4 * a * c));
  z = FN(clip01)(z);
  REAL x = bw * z + x0;
  REAL sq_z = z * z, z1mz = z - sq_z, sq_1mz = ((REAL)1 - z) * ((REAL)1 - z);
  REAL den = s + st * z1mz;
  REAL ld = -(REAL)2 * R_LOG(s) -
            R_LOG(d1 * sq_z + (REAL)2 * s * z1mz + d0 * sq_1mz) +
            (REAL)2 * R_LOG(den);
  if (y <= yk[0]) { x = (y - yk[0]) / dl[0] + xk[0]; ld = -R_LOG(dl[0]); }
  if (y >= yk[K]) { x = (y - yk[K]) / dl[K] + xk[K]; ld = -R_LOG(dl[K]); }
  *x_out = x;
  *ld_out = ld;
}

/* make_conditioner.conditioner, flows.py:46-86: input [c, v_0..v_{d-1}],
 * hk.nets.MLP(hidden, activate_final=True, relu) then hk.Linear(P). */
static void FN(conditioner)(const cnf_oracle_cfg *g, const REAL *p, int d,
                            REAL c, const REAL *v, REAL *theta) {
  int H = g->H, P = 3 * g->K + 1, nin = (g->periodized ? 2 : 1) * (1 + d);
  REAL h[CNF_ORACLE_MAX_H], h2[CNF_ORACLE_MAX_H];
  REAL in[2 * (CNF_ORACLE_MAX_D + 1)];
  const REAL *W = p, *b = p + (size_t)nin * H;
  if (g->periodized) {      /* flows.py:58-64: concatenate([sin(x)], [cos(x)]) of x = [c, v] */
    in[0] = R_SIN(c); in[1 + d] = R_COS(c);
    for (int i = 0; i < d; ++i) { in[1 + i] = R_SIN(v[i]); in[2 + d + i] = R_COS(v[i]); }
  } else {
    in[0] = c;
    for (int i = 0; i < d; ++i) in[1 + i] = v[i];
  }
  for (int j = 0; j < H; ++j) {
    REAL acc = b[j];
    /* (summed in the order c, v_0, v_1, ...: the non-periodized results stay bit-identical) */
    acc = b[j] + in[0] * W[j];
    for (int i = 1; i < nin; ++i) acc += in[i] * W[(size_t)i * H + j];
    h[j] = acc > 0 ? acc : (REAL)0;
  }
  p = b + H;
  for (int m = 1; m < g->M; ++m) {
    W = p; b = p + (size_t)H * H;
    for (int j = 0; j < H; ++j) {
      REAL acc = b[j];
      for (int i = 0; i < H; ++i) acc += h[i] * W[(size_t)i * H + j];
      h2[j] = acc > 0 ? acc : (REAL)0;
    }
    for (int j = 0; j < H; ++j) h[j] = h2[j];
    p = b + H;
  }
  W = p; b = p + (size_t)H * P;
  for (int j = 0; j < P; ++j) {
    REAL acc = b[j];
    for (int i = 0; i < H; ++i) acc += h[i] * W[(size_t)i * P + j];
    theta[j] = acc;
  }
}

static void FN(perm)(const cnf_oracle_cfg *g, int l, int *perm) {
  /* flows.py:141-143 with minimum_perm=True: arange(D), arange(D)[::-1], ... */
  for (int d = 0; d < g->D; ++d) perm[d] = (l % 2 == 0) ? d : g->D - 1 - d;
}

/* One sample, base -> data: flow.bijector.forward = ConditionalInverse(chain)
 * .forward = chain.inverse (conditional.py:169-177,233-237): layers 0..L-1,
 * each Autoregressive.inverse_and_log_det (autoregressive.py:109-136), which
 * conditions on the layer INPUT and uses the spline inverse. */
static void FN(forward1)(const cnf_oracle_cfg *g, const REAL *params,
                         const REAL *x, REAL c, REAL *y, REAL *logdet) {
  int D = g->D, K = g->K;
  REAL u[CNF_ORACLE_MAX_D], o[CNF_ORACLE_MAX_D], v[CNF_ORACLE_MAX_D];
  REAL theta[3 * CNF_ORACLE_MAX_K + 1];
  REAL xk[CNF_ORACLE_MAX_K + 1], yk[CNF_ORACLE_MAX_K + 1], dl[CNF_ORACLE_MAX_K + 1];
  int perm[CNF_ORACLE_MAX_D];
  REAL total = 0;
  for (int d = 0; d < D; ++d) u[d] = x[d];
  for (int l = 0; l < g->L; ++l) {
    FN(perm)(g, l, perm);
    for (int d = 0; d < D; ++d) {
      int i = perm[d];
      const REAL *th;
      if (d == 0) {
        th = params; /* shared `first`, ignores c: autoregressive.py:121-122 */
      } else {
        for (int q = 0; q < d; ++q) v[q] = u[perm[q]];
        FN(conditioner)(g, params + FN(cond_offset)(g, l, d), d, c, v, theta);
        th = theta;
      }
      FN(rqs_knots_bs)(th, K, (REAL)g->range_min, (REAL)g->range_max,
                       (REAL)g->min_bin_size, (REAL)g->min_knot_slope, g->periodized, xk, yk, dl);
      REAL ld;
      FN(rqs_inv)(u[i], xk, yk, dl, K, &o[i], &ld);
      total += ld;
    }
    for (int d = 0; d < D; ++d) u[d] = o[d];
  }
  for (int d = 0; d < D; ++d) y[d] = u[d];
  *logdet = total;
}

/* One sample, data -> base: flow.bijector.inverse = chain.forward
 * (conditional.py:159-167,239-243): layers L-1..0, each
 * Autoregressive.forward_and_log_det (autoregressive.py:76-107), which
 * conditions on already-transformed OUTPUTS and uses the spline forward. */
static void FN(inverse1)(const cnf_oracle_cfg *g, const REAL *params,
                         const REAL *y, REAL c, REAL *x, REAL *logdet) {
  int D = g->D, K = g->K;
  REAL u[CNF_ORACLE_MAX_D], o[CNF_ORACLE_MAX_D], v[CNF_ORACLE_MAX_D];
  REAL theta[3 * CNF_ORACLE_MAX_K + 1];
  REAL xk[CNF_ORACLE_MAX_K + 1], yk[CNF_ORACLE_MAX_K + 1], dl[CNF_ORACLE_MAX_K + 1];
  int perm[CNF_ORACLE_MAX_D];
  REAL total = 0;
  for (int d = 0; d < D; ++d) u[d] = y[d];
  for (int l = g->L - 1; l >= 0; --l) {
    FN(perm)(g, l, perm);
    for (int d = 0; d < D; ++d) {
      int i = perm[d];
      const REAL *th;
      if (d == 0) {
        th = params;
      } else {
        for (int q = 0; q < d; ++q) v[q] = o[perm[q]];
        FN(conditioner)(g, params + FN(cond_offset)(g, l, d), d, c, v, theta);
        th = theta;
      }
      FN(rqs_knots_bs)(th, K, (REAL)g->range_min, (REAL)g->range_max,
                       (REAL)g->min_bin_size, (REAL)g->min_knot_slope, g->periodized, xk, yk, dl);
      REAL ld;
      FN(rqs_fwd)(u[i], xk, yk, dl, K, &o[i], &ld);
      total += ld;
    }
    for (int d = 0; d < D; ++d) u[d] = o[d];
  }
  for (int d = 0; d < D; ++d) x[d] = u[d];
  *logdet = total;
}

static REAL FN(base_logprob)(const REAL *x, int D) {
  /* Independent(Normal(0,1)) log_prob, flows.py:166-173 */
  const REAL half_log_2pi = (REAL)0.91893853320467274178;
  REAL lp = 0;
  for (int d = 0; d < D; ++d) lp += -(REAL)0.5 * x[d] * x[d] - half_log_2pi;
  return lp;
}

/* ---- batch entry points --------------------------------------------------
 * c value of sample i is c[i / c_block] (c_block >= B: one uniform c;
 * c_block == 1: per-sample c, the vmap form of conditional.py:400). */

int FN(cnf_oracle_forward_logdet)(const cnf_oracle_cfg *g, const REAL *params,
                                  const REAL *x, const REAL *c, int64_t c_block,
                                  REAL *y, REAL *logdet, int64_t B) {
  if (cnf_oracle_check_cfg(g) != 0 || c_block < 1) return -1;
  int D = g->D;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < B; ++i) {
    REAL ld, yy[CNF_ORACLE_MAX_D];
    FN(forward1)(g, params, x + i * D, c[i / c_block], yy, &ld);
    for (int d = 0; d < D; ++d) y[i * D + d] = yy[d];
    if (logdet) logdet[i] = ld;
  }
  return 0;
}

int FN(cnf_oracle_inverse_logdet)(const cnf_oracle_cfg *g, const REAL *params,
                                  const REAL *y, const REAL *c, int64_t c_block,
                                  REAL *x, REAL *logdet, int64_t B) {
  if (cnf_oracle_check_cfg(g) != 0 || c_block < 1) return -1;
  int D = g->D;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < B; ++i) {
    REAL ld, xx[CNF_ORACLE_MAX_D];
    FN(inverse1)(g, params, y + i * D, c[i / c_block], xx, &ld);
    for (int d = 0; d < D; ++d) x[i * D + d] = xx[d];
    if (logdet) logdet[i] = ld;
  }
  return 0;
}

/* ConditionalTransformed.log_prob, conditional.py:316-321 */
int FN(cnf_oracle_log_prob)(const cnf_oracle_cfg *g, const REAL *params,
                            const REAL *value, const REAL *c, int64_t c_block,
                            REAL *logp, int64_t B) {
  if (cnf_oracle_check_cfg(g) != 0 || c_block < 1) return -1;
  int D = g->D;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < B; ++i) {
    REAL ld, xx[CNF_ORACLE_MAX_D];
    FN(inverse1)(g, params, value + i * D, c[i / c_block], xx, &ld);
    logp[i] = FN(base_logprob)(xx, D) + ld;
  }
  return 0;
}

/* ConditionalTransformed._sample_n_and_log_prob, conditional.py:382-402, with
 * the base draw x supplied by the caller (parity is defined on identical base
 * noise: SURVEY.md 7 "RNG parity"). */
int FN(cnf_oracle_sample_logprob)(const cnf_oracle_cfg *g, const REAL *params,
                                  const REAL *noise, const REAL *c,
                                  int64_t c_block, REAL *y, REAL *logp,
                                  int64_t B) {
  if (cnf_oracle_check_cfg(g) != 0 || c_block < 1) return -1;
  int D = g->D;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < B; ++i) {
    REAL ld, yy[CNF_ORACLE_MAX_D];
    FN(forward1)(g, params, noise + i * D, c[i / c_block], yy, &ld);
    for (int d = 0; d < D; ++d) y[i * D + d] = yy[d];
    if (logp) logp[i] = FN(base_logprob)(noise + i * D, D) - ld;
  }
  return 0;
}

/* Stand-alone scalar spline (arbitrary K / range / min slope): the object the
 * reference's tests/test_rqs_accuracy.py:71-77 exercises.  theta is [n][3K+1],
 * v is [n]; forward when inverse==0. */
int FN(cnf_oracle_rqs)(const REAL *theta, const REAL *v, int64_t n, int K,
                       REAL lo, REAL hi, REAL min_bin, REAL min_slope,
                       int inverse, REAL *out, REAL *logdet) {
  if (K < 1 || K > CNF_ORACLE_MAX_K || !(lo < hi)) return -1;
  for (int64_t i = 0; i < n; ++i) {
    REAL xk[CNF_ORACLE_MAX_K + 1], yk[CNF_ORACLE_MAX_K + 1], dl[CNF_ORACLE_MAX_K + 1];
    FN(rqs_knots)(theta + i * (3 * K + 1), K, lo, hi, min_bin, min_slope, xk, yk, dl);
    if (inverse) FN(rqs_inv)(v[i], xk, yk, dl, K, &out[i], &logdet[i]);
    else         FN(rqs_fwd)(v[i], xk, yk, dl, K, &out[i], &logdet[i]);
  }
  return 0;
}

/* knots only (for tests of the `first` table the HIP side precomputes) */
int FN(cnf_oracle_knots)(const REAL *theta, int K, REAL lo, REAL hi,
                         REAL min_bin, REAL min_slope, REAL *xk, REAL *yk,
                         REAL *dl) {
  if (K < 1 || K > CNF_ORACLE_MAX_K) return -1;
  FN(rqs_knots)(theta, K, lo, hi, min_bin, min_slope, xk, yk, dl);
  return 0;
}
