/*
 * cnf_ot_amd.h -- C ABI of the MI355X-native conditional RQS flow engine.
 *
 * This is the drop-in boundary for cnf_ot's flow-model call surface: the
 * namedtuple of pure functions returned by RQSFlow(...) and wrapped by
 * hk.multi_transform (reference: cnf_ot/models/flows.py:213-226, consumed by
 * cnf_ot/mfc/applications.py and cnf_ot/mfc/solvers.py:48-53).  Every entry
 * point below names the reference interface it replaces.  The reference is
 * pure Python (no FFI of its own): the binding a maintainer would add is the
 * ctypes stub shown in INTEGRATION.md; cnf_ot_amd/_capi.py is that stub.
 *
 * Conventions
 *  - plain C types only; all tensor arguments are DEVICE pointers to
 *    contiguous row-major float32 unless a parameter says otherwise;
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *    every compute entry point only enqueues work on that stream: no
 *    allocation, no synchronisation, graph-capturable.  Device memory is
 *    allocated by cnf_model_create, cnf_model_reserve and cnf_grad_enable only;
 *  - return value: 0 on success, negative CNF_ERR_* otherwise; nothing throws;
 *  - a CnfModel may be used from several host threads as long as each uses its
 *    own stream and nobody calls cnf_model_set_params concurrently; a compute
 *    call on another stream than the last cnf_model_set_params is ordered after
 *    it (event wait), but set_params does not wait for compute calls still in
 *    flight on OTHER streams -- finish those first.
 *
 * Condition argument (`c`, `c_block`): the condition of sample i is
 * c[i / c_block].  c_block == 1 is the per-sample form the reference uses for
 * sampling (cond[B,1] under vmap, conditional.py:400); c_block >= B is the
 * broadcast form it uses for log_prob (cond[1], autoregressive.py:96);
 * c_block == slice length fuses many time-slices into one launch (the shape of
 * cnf_ot/utils.py:311-340).
 */
#ifndef CNF_OT_AMD_H
#define CNF_OT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNF_OK 0
#define CNF_ERR_INVALID (-22)     /* bad argument / shape               */
#define CNF_ERR_UNSUPPORTED (-95) /* config has no compiled kernel      */
#define CNF_ERR_NOMEM (-12)
#define CNF_ERR_HIP (-5)          /* a HIP runtime call failed          */

/* Network configuration: RQSFlow(event_shape=(dim,), num_layers,
 * hidden_sizes=[hidden_size]*mlp_num_layers, num_bins) -- flows.py:178-199,
 * solvers.py:41-47; keys of config/mfc.yaml:29-33.  Spline constants are the
 * ones flows.py:124-132 passes to distrax.RationalQuadraticSpline. */
typedef struct CnfConfig {
  int32_t dim;            /* D: general.dim                         */
  int32_t num_layers;     /* L: cnf.flow_num_layers                 */
  int32_t hidden_size;    /* H: cnf.hidden_size                     */
  int32_t mlp_num_layers; /* M: cnf.mlp_num_layers (>= 1)           */
  int32_t num_bins;       /* K: cnf.num_bins                        */
  float range_min;        /* -10  (flows.py:127)                    */
  float range_max;        /* +10  (flows.py:128)                    */
  float min_bin_size;     /* 1e-4 (distrax default)                 */
  float min_knot_slope;   /* 1e-4 (flows.py:130)                    */
  int32_t periodized;     /* 0; 1 = RQSFlow(periodized=True), flows.py:58-64,127-131: the conditioner MLP sees
                           * [sin(x), cos(x)] of its input x = [c, v] (first linear layer: 2 (1 + d) rows, sin
                           * rows first), boundary_slopes='circular' (the last knot slope is the first); the
                           * caller sets range_min = 0, range_max = 2 pi (a float here: 6.2831855, 1.7e-7 above the reference's
                           * double).  Flow functions only (forward / inverse
                           * / log_prob / sample_logprob, float32 and float64): the loss, gradient and
                           * vector-Jacobian entry points return CNF_ERR_UNSUPPORTED -- no reference call site
                           * passes periodized=True. */
} CnfConfig;

typedef struct CnfModel CnfModel;

/* Fills the reference's defaults (config/mfc.yaml:29-33, flows.py:124-132). */
void cnf_config_default(CnfConfig *cfg, int32_t dim);

/* Number of float32 parameters of the flat layout below; < 0 on bad config.
 * 1 200 at dim=2, 11 824 at dim=10 (solvers.py:135-136 prints this count).
 *
 * Flat layout (this library's own order; it is NOT the order in which haiku
 * creates the parameters -- hk's init traces log_prob, which visits layer L-1
 * first (conditional.py:163-166) -- nor jax's key-sorted flattening: import a
 * haiku tree BY NAME, INTEGRATION.md / cnf_ot_amd/params.py from_tree):
 *   first[P]   then for l in 0..L-1, d in 1..D-1 (module names flows.py:46-86,146-158):
 *   mlp_layer{l}_d{d}/~/linear_0 w[(1+d)][H], b[H]; .../linear_{m} w[H][H],
 *   b[H] (m=1..M-1); linear_out_layer{l}_d{d} w[H][P], b[P];   P = 3K+1. */
int64_t cnf_param_count(const CnfConfig *cfg);

/* Replaces: RQSFlow(...) + hk.multi_transform (flows.py:178-226,
 * solvers.py:41-48).  Allocates the model's private device buffer (prepared
 * `first` knot table + a copy of the conditioner weights) on the current
 * device.  CNF_ERR_UNSUPPORTED if no kernel is compiled for (D,H,K). */
int cnf_model_create(const CnfConfig *cfg, CnfModel **out);
void cnf_model_destroy(CnfModel *m);

/* Replaces: passing `params` to model.apply.* (pure-function convention,
 * applications.py:85,153-158,233-239).  `params` is a device pointer to
 * cnf_param_count() float32 in the flat layout; enqueues a small kernel that
 * normalises the shared, condition-independent `first` spline once (float64
 * on device) and snapshots the weights.  Must precede compute calls; call
 * again after every optimiser step. */
int cnf_model_set_params(CnfModel *m, const float *params, void *stream);

/* Workspace of the dim-2 table path (no reference counterpart: the reference
 * re-evaluates the conditioner MLP per sample, flows.py:57-84).  Large dim-2
 * calls whose condition is uniform over slices read the conditioner from exact
 * piecewise-linear tables, one SET (num_layers tables, cnf_model_table_bytes()
 * bytes) per time-slice and condition; the fused loss terms need up to three
 * sets per slice (t - dt/2, t + dt/2, t).  cnf_model_reserve(m, stream, n_sets)
 * makes room for n_sets sets for calls on `stream` (grows only; n_sets = 0
 * releases it).  It allocates: call it outside graph capture.  A block that a
 * larger reservation replaces stays allocated until cnf_model_destroy (or the
 * release), so a graph captured on the smaller one can still be replayed; the
 * release (n_sets = 0) synchronises `stream`, frees everything and invalidates
 * graphs captured on this model's table path.  Compute calls never allocate: a call with more slices than
 * the reservation holds is processed in chunks, and without a (useful)
 * reservation the same result comes from the MLP kernels.  2 048 sets cover the
 * largest chunk a call is ever split into. */
int cnf_model_reserve(CnfModel *m, void *stream, int64_t n_sets);
int64_t cnf_model_reserved(CnfModel *m, void *stream);
int64_t cnf_model_table_bytes(const CnfModel *m);

/* Numerics of the data -> base direction (cnf_log_prob, cnf_inverse_logdet).
 * log_prob = sum_d -x_d^2/2 + ildj multiplies the error of the recovered base
 * point by |x| (up to 5), and an all-fp32 evaluation of the softmax-normalised
 * knots leaves ~2e-6 in x: max |d log_prob| 1.1e-5 .. 1.5e-5 against a float64
 * evaluation of the reference algorithm (conditional.py:316-321 in float64,
 * solvers.py:23) on 65 536 samples.  on = 1 (the default): the softmax terms are
 * evaluated to ~1e-9, and the knot prefix sums, the bin corner, the offset in
 * the bin, the result and the base term are carried in float64 (the conditioner
 * MLP, the slopes and the log-det terms stay fp32): max |d log_prob| ~2e-6, at
 * ~1.5x the time of a data -> base call.  on = 0: plain fp32 throughout (what
 * the fused loss terms use internally for their score differences). */
int cnf_model_set_precise(CnfModel *m, int on);

/* Which kernels the most recent compute call on this model ran: 1/2 = MLP flow
 * kernel with one / two samples per lane, 3 = MFMA conditioner, 4 = conditioner
 * tables, 5/6 = fused loss kernel on the MLP / on tables, 7 = float64. */
int cnf_model_last_path(const CnfModel *m);

/* Replaces: model.apply.forward(params, x, c) = flow.bijector.forward, and
 * flow.bijector.forward_and_log_det (flows.py:221-223, conditional.py:233-237,
 * 169-177; autoregressive.py:109-136).  base -> data.  y [B,D]; logdet [B]
 * may be NULL. */
int cnf_forward_logdet(CnfModel *m, const float *x, const float *c,
                       int64_t c_block, float *y, float *logdet, int64_t B,
                       void *stream);

/* Replaces: model.apply.inverse(params, y, c) = flow.bijector.inverse /
 * inverse_and_log_det (conditional.py:239-243,159-167;
 * autoregressive.py:76-107).  data -> base.  logdet may be NULL. */
int cnf_inverse_logdet(CnfModel *m, const float *y, const float *c,
                       int64_t c_block, float *x, float *logdet, int64_t B,
                       void *stream);

/* Replaces: model.apply.log_prob(params, value, cond)
 * (conditional.py:316-321; call sites applications.py:85,272). */
int cnf_log_prob(CnfModel *m, const float *value, const float *c,
                 int64_t c_block, float *logp, int64_t B, void *stream);

/* Replaces: model.apply.sample / sample_and_log_prob(params, cond=, seed=,
 * sample_shape=(B,)) (conditional.py:323-402; call sites
 * applications.py:153-158,233-239, utils.py:330-336) for base noise the caller
 * supplies (noise [B,D] ~ N(0,I), e.g. from cnf_fill_normal).  y [B,D];
 * logp [B] may be NULL (= sample).  THE METRIC KERNEL of BASELINE.json. */
int cnf_sample_logprob(CnfModel *m, const float *noise, const float *c,
                       int64_t c_block, float *y, float *logp, int64_t B,
                       void *stream);

/* Replaces: model.apply.sample / sample_and_log_prob(params, cond=, seed=, sample_shape=(B,)) as the reference
 * calls them (conditional.py:376-402: the base draw happens INSIDE the call): the same as cnf_fill_normal +
 * cnf_sample_logprob, bit for bit, with the noise drawn in the flow kernel -- no noise tensor in HBM (12 instead
 * of 20 bytes per sample at dim 2) and one launch instead of two.  Sample i of the call is sample
 *   first_sample + (i / c_block) * slice_stride + i % c_block
 * of the cnf_fill_normal stream of `seed` (element (sample) * D + d): slice_stride = c_block (or one slice,
 * c_block >= B) draws B consecutive samples, slice_stride = 0 gives every slice the same draw (the reused rng of
 * applications.py:392-400).  Per-sample conditions (c_block = 1) need slice_stride = 1.  logp may be NULL. */
int cnf_sample_logprob_seeded(CnfModel *m, uint64_t seed, int64_t first_sample,
                              int64_t slice_stride, const float *c,
                              int64_t c_block, float *y, float *logp, int64_t B,
                              void *stream);

/* float64 instantiation of the four functions above: the reference computes in
 * float64 (jax.config.update("jax_enable_x64", True), solvers.py:23).  Double
 * IO, double spline table and constants, ocml math, one sample per lane:
 * agrees with a float64 evaluation of the reference algorithm to ~1e-12, at a
 * fraction of the fp32 kernels' speed.  The parameters are the float32 vector
 * given to cnf_model_set_params (each weight is widened exactly). */
int cnf_forward_logdet_f64(CnfModel *m, const double *x, const double *c,
                           int64_t c_block, double *y, double *logdet, int64_t B,
                           void *stream);
int cnf_inverse_logdet_f64(CnfModel *m, const double *y, const double *c,
                           int64_t c_block, double *x, double *logdet, int64_t B,
                           void *stream);
int cnf_log_prob_f64(CnfModel *m, const double *value, const double *c,
                     int64_t c_block, double *logp, int64_t B, void *stream);
int cnf_sample_logprob_f64(CnfModel *m, const double *noise, const double *c,
                           int64_t c_block, double *y, double *logp, int64_t B,
                           void *stream);

/* Replaces: the base draw `Independent(Normal(0,1)).sample(seed=rng, B)`
 * (conditional.py:378,399).  JAX's threefry stream cannot be reproduced
 * (JAX absent, stream version-dependent); the build's stream is
 * Philox4x32-10 + Box-Muller, a pure function of (seed, element index):
 * out[i] = normal(seed, first_element + i).  Sharding a batch over GPUs with
 * first_element = rank_offset * D gives results independent of the GPU count. */
int cnf_fill_normal(uint64_t seed, uint64_t first_element, int64_t n,
                    float *out, void *stream);

/* The same draw as JAX makes it: jax.random.normal(key, shape, float64) with
 * prod(shape) = `size`, key data (key0, key1) (jax.random.PRNGKey(seed) is
 * (seed >> 32, seed & 0xffffffff)), classic (non-partitionable) threefry bit
 * generation; elements [first_element, first_element + n) of the flattened
 * draw, as float32 and / or float64 (either pointer may be NULL).  Threefry-2x32
 * is pinned by the Random123 known answers; the bits -> normal mapping restates
 * jax._src.random and could not be compared with JAX itself (not installable
 * where this was built): use it to reproduce a reference run's base noise, and
 * verify on a JAX box before relying on bit-level agreement. */
int cnf_fill_normal_threefry(uint32_t key0, uint32_t key1, uint64_t size,
                             uint64_t first_element, int64_t n, float *out_f32,
                             double *out_f64, void *stream);

/* ---- fused Monte-Carlo loss terms (cnf_ot/mfc/applications.py) ------------
 * One launch evaluates one term over n_slices time-slices x B samples and
 * returns, per slice, the SUM over that slice's samples (double, device); the
 * caller divides by the global sample count and composes the losses
 * (applications.py:377-441), after an all-reduce when samples are sharded.
 * No [B,D] intermediate reaches HBM. */
enum CnfTermKind {
  /* kinetic_loss_fn, applications.py:220-242 (and utils.py:311-340):
   * sum_{i,d} ((F(x_i,t+dt/2) - F(x_i,t-dt/2)) / dt)^2 on the SAME base noise */
  CNF_TERM_KINETIC = 0,
  /* kinetic_with_score_loss_fn, applications.py:245-276 (utils.py:343-389):
   * v = (r2-r1)/dt + coef * score(r3), score_d by central differences of
   * log_prob at r3 +- dx/2 e_d; sum_{i,d} v^2; coef = 1/beta */
  CNF_TERM_KINETIC_SCORE = 1,
  /* flow_matching_loss_fn, applications.py:279-374: as above with coef = sigma
   * and sum_{i,d} (v - drift_d(r3))^2; drift by `subtype` */
  CNF_TERM_FLOW_MATCHING = 2,
  /* potential_loss_fn, applications.py:176-205: sum_i V(F(x_i,t)) */
  CNF_TERM_POTENTIAL = 3,
  /* reverse_kl_loss_fn, applications.py:129-163:
   * sum_i log_prob_i - log(N(y_i;0,2/beta (T+1) I)(T-t)/T + N(y_i;0,2/beta I) t/T) */
  CNF_TERM_REVERSE_KL = 4,
  /* kl_loss_fn, applications.py:11-86 (after the host mixed the samples):
   * sum_i -log_prob(value_i; t); pts are DATA points, not base noise */
  CNF_TERM_NEG_LOGPROB = 5
};
enum CnfPotential { CNF_POT_QUADRATIC = 0, CNF_POT_DOUBLE_WELL = 1, CNF_POT_OBSTACLE = 2 };
/* drift of flow_matching_loss_fn: OU = -a r (applications.py:310, README);
 * SMILE = the 2-D field that overwrites it (applications.py:353-357);
 * NONGRADIENT (:358-363, dim 2); LORENZ (:364-372, dim 3) */
enum CnfDrift { CNF_DRIFT_OU = 0, CNF_DRIFT_SMILE = 1, CNF_DRIFT_NONGRADIENT = 2, CNF_DRIFT_LORENZ = 3 };

typedef struct CnfLossSpec {
  int32_t kind;      /* CnfTermKind                                         */
  int32_t subtype;   /* CnfPotential or CnfDrift                            */
  float dt, dx;      /* finite-difference steps (general.dt / general.dx)   */
  float coef;        /* 1/beta (KINETIC_SCORE) or sigma (FLOW_MATCHING)     */
  float a;           /* potential / drift parameter (rwpo.a, fp.a)          */
  float T, beta;     /* REVERSE_KL                                          */
} CnfLossSpec;

/* pts: [n_slices * B, D] when pts_shared == 0 (each slice its own draw, the
 * key-split of utils.py:328), [B, D] when pts_shared != 0 (every slice reuses
 * the same draw: the reused rng of applications.py:392-400).  t: [n_slices].
 * sums: [n_slices] doubles, overwritten. */
int cnf_loss_terms(CnfModel *m, const CnfLossSpec *spec, const float *pts,
                   int pts_shared, const float *t, int64_t n_slices, int64_t B,
                   double *sums, void *stream);

/* Same as cnf_loss_terms with the base noise drawn inside the kernel (no noise
 * tensor in HBM): sample i of slice s is sample (first_sample + s * slice_stride
 * + i) of the cnf_fill_normal stream of `seed`.  slice_stride = 0: every slice
 * reuses the same draw (applications.py:392-400); slice_stride = batch size:
 * each slice its own draw (the key split of utils.py:328).  Not for
 * CNF_TERM_NEG_LOGPROB (whose points are data). */
int cnf_loss_terms_seeded(CnfModel *m, const CnfLossSpec *spec, uint64_t seed,
                          int64_t first_sample, int64_t slice_stride,
                          const float *t, int64_t n_slices, int64_t B,
                          double *sums, void *stream);

/* ---- value_and_grad + Adam (cnf_ot/mfc/solvers.py:90-97) -------------------
 * Backward pass of the loss terms, for the reference's network (hidden 16, two
 * hidden layers, 5 bins; dim <= 14 -- the first layer's inputs + bias row are 16 MFMA rows, and the tile's working set
 * must fit one CU's LDS): cnf_grad_supported() tells.  A model's gradient
 * slabs are shared state: do not run two gradient calls of the same model
 * concurrently on different streams.
 *
 * cnf_grad_enable allocates the per-wave gradient slabs (the only allocation;
 * call once, outside any graph capture; max_blocks <= 0: a default).
 *
 * cnf_loss_terms_grad = cnf_loss_terms (same arguments, same `sums`) PLUS
 *   grad[p] += scale * d(sum over all slices and samples of the term)/d params[p]
 * `grad` (device, cnf_param_count() floats) is ACCUMULATED into, so the caller
 * zeroes it once and adds every term of a composite loss with its coefficient
 * as `scale`.  `params` is the same flat vector that was given to
 * cnf_model_set_params.  Results are deterministic (no float atomics). */
int cnf_grad_supported(const CnfConfig *cfg);
int cnf_grad_enable(CnfModel *m, int64_t max_blocks);
int cnf_loss_terms_grad(CnfModel *m, const CnfLossSpec *spec, const float *pts,
                        int pts_shared, const float *t, int64_t n_slices,
                        int64_t B, float scale, double *sums, float *grad,
                        const float *params, void *stream);

/* Up to 4 terms of one composite loss (applications.py:377-441) in ONE launch: term i with its own points, slices,
 * batch and coefficient (arrays of n_terms entries; sums[i] has n_slices[i] doubles).  The terms' tiles share the grid,
 * so several small terms run side by side instead of one under-filled launch after the other -- a default-config
 * training step (config/mfc.yaml: batch 2 048) is three terms of 8 tiles each.  Same result as n_terms calls of
 * cnf_loss_terms_grad up to the order of the float32 additions into `grad`. */
int cnf_loss_terms_grad_multi(CnfModel *m, int32_t n_terms, const CnfLossSpec *specs,
                              const float *const *pts, const int32_t *pts_shared,
                              const float *const *t, const int64_t *n_slices,
                              const int64_t *B, const float *scale,
                              double *const *sums, float *grad, const float *params,
                              void *stream);

/* Replaces the autodiff helpers of the Flow tuple (flows.py:203-211):
 *   forward_jac = vmap(jacfwd(flow.bijector.forward)),
 *   inverse_jac = vmap(jacfwd(flow.bijector.inverse)),
 *   gauge_potential = jacfwd(log|det J| of forward)
 * through one primitive, the vector-Jacobian product of a flow pass w.r.t. its
 * input points:  xbar[b,:] = ybar[b,:] . dF/dx(b) + ldbar[b] * d logdet/dx(b).
 * to_base = 0: F = flow.bijector.forward (base -> data); 1: the inverse.
 * ybar [B,D] and ldbar [B] may each be NULL (= 0), not both.  Row i of the
 * Jacobian is the call with ybar = e_i.  Same config support as the gradients. */
int cnf_input_vjp(CnfModel *m, int to_base, const float *pts, const float *c,
                  int64_t c_block, const float *ybar, const float *ldbar,
                  float *xbar, int64_t B, void *stream);

/* The backward of a differentiable flow op: cnf_input_vjp PLUS the parameter
 * gradient of the same pass,
 *   grad[p] += sum_b ( ybar[b,:] . dF/dp(b) + ldbar[b] * d logdet/dp(b) ),
 * so that any loss composed on the host from flow passes (e.g. under
 * torch.autograd: cnf_ot_amd/autograd.py) gets exact gradients -- what
 * jax.value_and_grad gives the reference for losses not in applications.py.
 * xbar may be NULL; grad is accumulated (needs cnf_grad_enable).
 * At dim 2, with a condition that is uniform over slices (c_block >= 8 192) and >= 262 144 points, and tables
 * reserved on the stream for min(n_slices, 128) slices (cnf_model_reserve), the pass runs on the conditioner tables:
 * per-piece sufficient statistics in 64-bit fixed point instead of per-sample weight gradients --
 * same result to ~1e-6, about twice as fast, bitwise reproducible.  Nothing is allocated here: the statistics
 * buffer comes with cnf_grad_enable. */
int cnf_pass_vjp(CnfModel *m, int to_base, const float *pts, const float *c,
                 int64_t c_block, const float *ybar, const float *ldbar,
                 float *xbar, float *grad, const float *params, int64_t B,
                 void *stream);

/* Value and gradient of the density-fit term in ONE launch over the data (kl_loss_fn, applications.py:11-86, under
 * jax.value_and_grad, solvers.py:94):
 *   sums[s] = -sum_{i in slice s} log_prob(pts_i; c_s),   grad[p] += loss_coef * d(sum_s sums[s]) / dp
 * -- the table form of cnf_pass_vjp in the data -> base direction with the output adjoints formed in the kernel
 * (ybar = loss_coef * base point, ldbar = -loss_coef), so neither cnf_inverse_logdet nor cnf_term_residual nor the
 * adjoint scan run, and the tables are built once.  Same conditions as the table form of cnf_pass_vjp;
 * CNF_ERR_UNSUPPORTED where it does not apply: the caller composes the term from those three calls. */
int cnf_neg_logprob_vjp(CnfModel *m, const float *pts, const float *c,
                        int64_t c_block, float loss_coef, double *sums,
                        float *grad, const float *params, int64_t B,
                        void *stream);

/* Value and gradient of the kinetic (+ potential) term of ot_loss_fn in one call (kinetic_loss_fn / potential_loss_fn,
 * applications.py:176-242, as ot_loss_fn combines them, :388-402, under jax.value_and_grad, solvers.py:94): the ONE
 * base draw z [count, 2] pushed to the S times t_s - dt/2, t_s + dt/2 and -- with subtype >= 0 (CnfPotential) -- t_s:
 *   kin[s] = sum_i |(r2_i - r1_i) / dt|^2,  pot[s] = sum_i V(r3_i),
 *   grad[p] += c_kin d(sum_s kin[s]) / dp + c_pot d(sum_s pot[s]) / dp.
 * c: the 2 S (3 S) conditions [t - dt/2 | t + dt/2 | t] as the caller rounds them.  One table build, one forward
 * launch in which all slices read the same z (no repeated copy of it), the term epilogues, which also leave the largest
 * adjoint for the backward (no scan), and one backward launch -- what cnf_sample + cnf_term_residual + cnf_pass_vjp do
 * in six launches over 2 (3) S repeated copies of z.  work: 4 x 2 (3) S x count floats, 16-byte aligned (the pushed
 * points and their adjoints).  pot == NULL iff subtype < 0.  grad == NULL: the terms' values alone (the loss without
 * jax.value_and_grad; work: 2 x 2 (3) S x count floats).  CNF_ERR_UNSUPPORTED (nothing written) where the table
 * backward does not apply or 2 (3) S > 128: compose the term from those calls. */
int cnf_kinetic_potential_vjp(CnfModel *m, const float *z, int64_t count,
                              const float *c, int32_t S, float dt, float c_kin,
                              int32_t subtype, float pot_a, float c_pot,
                              double *kin, double *pot, float *grad,
                              const float *params, float *work, void *stream);

/* The score of the flow's density by central differences, the way the reference
 * forms it (kinetic_with_score_loss_fn / flow_matching_loss_fn,
 * applications.py:264-273; utils.py:366-381):
 *   score[i, d] = (log_prob(r_i + dx/2 e_d) - log_prob(r_i - dx/2 e_d)) / dx
 * for B points r_i [B, D] (the 2 D evaluation points of a point are generated in
 * the kernel: 2 D B flow passes spread over the whole GPU, nothing but `score`
 * [B, D] is written).  cnf_logprob_fd_vjp is its backward:
 *   pts_bar[i, :] = sum_d gbar[i, d] * d score[i, d] / d r_i      (may be NULL)
 *   grad[p]      += sum_{i,d} gbar[i, d] * d score[i, d] / d params[p]
 * (needs cnf_grad_enable; same config support as the other gradients). */
int cnf_logprob_fd(CnfModel *m, const float *pts, const float *c,
                   int64_t c_block, float dx, float *score, int64_t B,
                   void *stream);
int cnf_logprob_fd_vjp(CnfModel *m, const float *pts, const float *c,
                       int64_t c_block, float dx, const float *gbar,
                       float *pts_bar, float *grad, const float *params,
                       int64_t B, void *stream);

/* The score terms' value AND backward in one launch (value_and_grad of kinetic_with_score_loss_fn /
 * flow_matching_loss_fn, applications.py:245-374, composed from separate flow launches -- the form
 * cnf_ot_amd.applications uses from dim 6 up).  r [3n, D] = the samples at t - dt/2 | t + dt/2 | t from one
 * base -> data launch; per slice of `count` points (condition c[slice])
 *   sums[s] = sum_{i,d} u_{i,d}^2,  u = (r2 - r1)/dt + coef * score_d(r3) - drift_d(r3),
 * score by central differences of log_prob as in cnf_logprob_fd, drift = CNF_DRIFT_OU (-a r) or -1 (none).
 * For d(loss) = loss_coef * d(sum of sums):  rbar [3n, D] receives the adjoints of r (all three blocks, the
 * r3 block complete: drift, score and the 2 D evaluation points' input adjoints), grad the parameter gradient
 * (accumulated; needs cnf_grad_enable).  Equals cnf_logprob_fd + cnf_score_residual + cnf_logprob_fd_vjp, but
 * the 2 D n evaluation points are pushed through the flow ONCE: the kernel that differentiates them forms the
 * score from its own forward passes (no separate forward launch, no score / sbar tensors). */
int cnf_score_fd_vjp(CnfModel *m, const float *r, const float *c, int64_t count,
                     float dt, float dx, float coef, int32_t drift, float a,
                     float loss_coef, double *sums, float *rbar, float *grad,
                     const float *params, int64_t n, void *stream);

/* Epilogues of loss terms composed from separate flow launches (what
 * cnf_ot_amd.applications does from dim 6 up, where a rank's few samples cannot
 * fill the GPU from inside one fused kernel).  Model-independent, like
 * cnf_adam_step.
 *
 * cnf_score_residual: r [3n, D] = the samples at t - dt/2 | t + dt/2 | t (one
 * base -> data launch), score [n, D] from cnf_logprob_fd; per slice of `count`
 * samples  sums[s] = sum_{i,d} ((r2 - r1)/dt + coef score - drift_d(r3))^2
 * (kinetic_with_score_loss_fn / flow_matching_loss_fn, applications.py:245-374;
 * drift = CnfDrift, or -1 for none).  With rbar / sbar non-NULL it also writes
 * the adjoints of r and score for d(loss) = loss_coef * d(sum of sums): the
 * seeds of cnf_logprob_fd_vjp and cnf_pass_vjp.
 *
 * cnf_rkl_residual: y [n, D], lp [n] from cnf_sample_logprob;
 * sum = sum_i lp_i - log(mixture(y_i)) of reverse_kl_loss_fn
 * (applications.py:129-163), adjoints ybar / lpbar likewise. */
/* cnf_term_residual: value and adjoints of the kinetic / potential / density-fit terms composed from separate flow
 * launches (kinetic_loss_fn applications.py:220-242: r = [r1 | r2], 2 n points, p0 = dt; potential_loss_fn :176-205:
 * r = n points, subtype = CnfPotential, p0 = a; kl_loss_fn :11-86: r = the recovered base points, aux = ildj).
 * sums [ceil(n / count)] per slice; rbar (and auxbar for the density fit) receive loss_coef * d(sum) / d(.) when
 * non-NULL. */
int cnf_term_residual(int32_t kind, const float *r, const float *aux, int64_t n,
                      int64_t count, int32_t D, int32_t subtype, float p0,
                      float loss_coef, double *sums, float *rbar, float *auxbar,
                      void *stream);
int cnf_score_residual(const float *r, const float *score, int64_t n,
                       int64_t count, int32_t D, float dt, float coef,
                       int32_t drift, float a, float loss_coef, double *sums,
                       float *rbar, float *sbar, void *stream);
int cnf_rkl_residual(const float *y, const float *lp, int64_t n, int32_t D,
                     float t, float T, float beta, float loss_coef, double *sum,
                     float *ybar, float *lpbar, void *stream);

/* optax.adam(lr) update in place (solvers.py:55,95-96): b1 = 0.9, b2 = 0.999,
 * eps = 1e-8 are optax's defaults; `step` counts from 1. */
int cnf_adam_step(float *params, const float *grad, float *mu, float *nu,
                  int64_t n, float lr, float b1, float b2, float eps,
                  int64_t step, void *stream);

/* ---- a training step as ONE device-side program (cnf_ot/mfc/solvers.py:90-105: the reference's step is one jitted
 * XLA program; here: one HIP graph, captured once and replayed) ----------------------------------------------------
 * What changes from step to step must not be a kernel argument of a captured step: the random key and the step
 * count live in device memory, `state` = uint64[2] = { step count, key }.  The caller writes state[1] (one 8-byte
 * copy) before a step; cnf_step_begin increments the count; the draws below read the key on the device:
 *   cnf_fill_normal_dev     out[i] = element first_element + i of the cnf_fill_normal stream of the key
 *                           (base noise; conditional.py:378,399)
 *   cnf_fill_uniform_dev    out[i] = scale * uniform[0, 1)        (the time batch, applications.py:392,414,434)
 *   cnf_mixture_source_dev  out[i, :] = z[i, :] + centre of a uniformly drawn component of the 8-mode source
 *                           (dim 2; applications.py:34-71); comp (optional) receives the component indices
 * cnf_adam_step_dev is cnf_adam_step with `step` = state[0]; cnf_weighted_sum (out = sum v[i] w[i], one block, fixed
 * order) composes a loss from its terms' partial sums without a BLAS call inside the capture. */
int cnf_step_begin(uint64_t *state, void *stream);
int cnf_fill_normal_dev(const uint64_t *state, uint64_t first_element, int64_t n,
                        float *out, void *stream);
int cnf_fill_uniform_dev(const uint64_t *state, uint64_t first, int64_t n,
                         float scale, float *out, void *stream);
int cnf_mixture_source_dev(const uint64_t *state, uint64_t first_sample, int64_t n,
                           const float *z, float *out, int32_t *comp, void *stream);
int cnf_adam_step_dev(float *params, const float *grad, float *mu, float *nu,
                      int64_t n, float lr, float b1, float b2, float eps,
                      const uint64_t *state, void *stream);
int cnf_weighted_sum(const double *v, const double *w, int64_t n, double *out,
                     void *stream);

const char *cnf_strerror(int code);
/* "gfx950" etc.: the offload arch this library was compiled for. */
const char *cnf_build_arch(void);
/* 1 if a kernel is compiled for this (dim, hidden_size, num_bins). */
int cnf_config_supported(const CnfConfig *cfg);

#ifdef __cplusplus
}
#endif
#endif /* CNF_OT_AMD_H */
