/*
 * cnf_ot_amd_debug.h -- test and measurement knobs of libcnf_ot_amd.so that are NOT part of the drop-in boundary
 * (include/cnf_ot_amd.h): kernel-path selection for the parity tests (every path is checked against the oracle on
 * its own) and the HIP-event profile bench.py's `roofline` object reads.  A cnf_ot integration never calls these;
 * the defaults are what the library ships with.  cnf_ot_amd/_capi.py binds them as `_INTERNAL`.
 */
#ifndef CNF_OT_AMD_DEBUG_H
#define CNF_OT_AMD_DEBUG_H

#include "cnf_ot_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 1 (default): hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32 / v_sqrt_f32); 0: ocml's functions.
 * Both meet the same tolerances (tests/test_gpu_parity.py). */
int cnf_model_set_fast_math(CnfModel *m, int on);

/* 0 (default): by batch size; 1 / 2: force the one-sample-per-lane or the packed two-samples-per-lane MLP kernel. */
int cnf_model_set_samples_per_lane(CnfModel *m, int spl);

/* MFMA (v_mfma_f32_16x16x4_f32) conditioner of the flow kernels: 2 (default) for launches that leave the chip
 * under-filled, 1 wherever available, 0 never. */
int cnf_model_set_mfma(CnfModel *m, int mode);

/* dim-2 conditioner tables: 1 (default) for large launches, 2 whenever they apply, 0 never. */
int cnf_model_set_pwl(CnfModel *m, int mode);

/* wave-per-dimension kernel (base -> data, dim >= 3): 1 (default) by batch size, 2 always, 0 never. */
int cnf_model_set_dpar(CnfModel *m, int mode);

/* on = 1: the flow entry points record HIP events on their launch stream around each kernel they enqueue (table
 * path: before the table build, between build and flow kernel, after the flow kernel).  cnf_model_read_profile
 * waits for the recorded launches and returns, summed since the last read: the flow kernels' and the table
 * builder's milliseconds, the number of (flow) launches and the samples they processed. */
int cnf_model_set_profiling(CnfModel *m, int on);
int cnf_model_read_profile(CnfModel *m, double *flow_ms, double *build_ms,
                           int64_t *launches, int64_t *samples);

#ifdef __cplusplus
}
#endif
#endif /* CNF_OT_AMD_DEBUG_H */
