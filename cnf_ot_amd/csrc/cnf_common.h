// cnf_common.h -- structs shared by the translation units of libcnf_ot_amd.so
// (cnf_flow.hip: forward kernels + C ABI; cnf_grad.hip: backward + Adam).
#pragma once
#include <mutex>
#include <map>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "cnf_device.h"
#include "../../include/cnf_ot_amd.h"

namespace cnf {

constexpr int TILE = 256;
constexpr double HALF_LOG_2PI = 0.91893853320467274178;

struct ModelArgs {
  const float* prep;     // prepared model buffer
  const float* wq;       // MFMA-layout weights (inside prep), or null
  int64_t per_layer;     // floats of conditioner weights per flow layer
  int64_t per_layer_q;   // floats of MFMA-layout weights per flow layer
  int32_t D, L, M;
  SplineConsts sc;
  SplineConstsT<double> scd;   // the same constants in float64 (exact-mode kernels)
  const double* tabd;          // float64 copy of the `first` table (inside prep)
  const double* e2tab;         // 2^(-i/32), i = 0 .. 1024 (inside prep): the precise position path
};

template <class R> __device__ __forceinline__ const SplineConstsT<R>& sc_of(const ModelArgs& a);
template <> __device__ __forceinline__ const SplineConstsT<float>& sc_of<float>(const ModelArgs& a) { return a.sc; }
template <> __device__ __forceinline__ const SplineConstsT<double>& sc_of<double>(const ModelArgs& a) { return a.scd; }
template <class R> __device__ __forceinline__ const R* table_of(const ModelArgs& a);
template <> __device__ __forceinline__ const float* table_of<float>(const ModelArgs& a) { return a.prep; }
template <> __device__ __forceinline__ const double* table_of<double>(const ModelArgs& a) { return a.tabd; }

// target drift of flow_matching_loss_fn at r (this thread's column `r3`), dim i
template <class T>
__device__ __forceinline__ T drift_of(const float* r3, int i, int D, int TS, int subtype, float a) {
  const T ri = lds_get<T>(r3, i, TS);
  switch (subtype) {
    case CNF_DRIFT_SMILE: {          // applications.py:353-357 (2-D)
      const T x = lds_get<T>(r3, 0, TS), y = lds_get<T>(r3, 1, TS);
      const T q = x * x + y * y - 4.0f;
      return (i == 0 ? -q * x : -q * y - (y - 1.0f) * 2.0f) * a;
    }
    case CNF_DRIFT_NONGRADIENT: {    // applications.py:358-363: -a r + 0.5 (r @ J), J=[[0,1],[-1,0]]
      const T x = lds_get<T>(r3, 0, TS), y = lds_get<T>(r3, 1, TS);
      return i == 0 ? x * -a - y * 0.5f : y * -a + x * 0.5f;
    }
    case CNF_DRIFT_LORENZ: {         // applications.py:364-372, _r = 9
      const T x = lds_get<T>(r3, 0, TS), y = lds_get<T>(r3, 1, TS), z = lds_get<T>(r3, 2, TS);
      if (i == 0) return (y - x) * 10.0f;
      if (i == 1) return x * 9.0f * (splat<T>(28.0f / 9.0f) - z) - y;
      return x * 9.0f * y - z * (8.0f / 3.0f);
    }
    default: return ri * -a;         // OU drift, applications.py:310
  }
}

}  // namespace cnf

struct CnfModel {
  // (opaque to callers; see include/cnf_ot_amd.h)
  CnfConfig cfg;
  cnf::SplineConsts sc;
  cnf::SplineConstsT<double> scd;
  int64_t tabd_off;       // offset (in floats, 8-byte aligned) of the float64 table inside prep
  int64_t e2_off;         // offset (in floats, 8-byte aligned) of the 2^(-i/32) table inside prep
  int precise;            // 1 (default): data -> base entry points run the precise position path
  float* prep;            // device
  int64_t n_params;
  int64_t per_layer;
  int device;
  int num_cus;
  int fast_math;          // 1: hardware transcendentals (default), 0: ocml
  int force_spl;          // 0: automatic; 1 / 2: samples per lane (tests, bench)
  int use_mfma;           // 1: MFMA conditioner where available (H = 16, K = 5, fast math)
  uint32_t div_magic;     // ceil(2^32 / D)
  int64_t per_layer_q;    // MFMA-layout floats per flow layer (0: not available)
  int64_t mfma_off;       // offset of the MFMA-layout weights inside prep (floats)
  int params_set;
  float* grad_slabs;      // per-wave gradient slabs (cnf_grad_enable), or null
  int64_t grad_max_blocks;
  int use_pwl;            // 1: piecewise-linear conditioner tables at dim 2 (cnf_pwl.h)
  int use_dpar;           // 1: wave-per-dimension kernel for small base -> data launches at dim >= 3
  // table workspaces [sets][L][PWL_TBL], one per stream (calls on different streams never share one).
  // Allocated ONLY by cnf_model_reserve; the compute entry points look theirs up and never allocate.
  // cnf_grad_enable: per-piece gradient statistics of the table backward (cnf_grad.hip), PWL_STAT_SLICES slices
  float* pwl_stats;
  // retired: blocks of this stream outgrown by a later reservation, kept alive for graphs captured on them
  struct PwlWorkspace { float* tables; int64_t sets; uint32_t epoch; std::vector<float*> retired; };
  std::mutex pwl_mu;
  std::unordered_map<void*, PwlWorkspace> pwl_ws;
  // cnf_model_set_params records `prep_event` after its kernel; a compute call on another stream waits for it
  hipEvent_t prep_event;
  void* prep_stream;
  int last_path;          // CnfPath of the most recent compute call (cnf_model_last_path)
  // cnf_model_set_profiling: HIP events around the kernels of the flow entry points
  struct ProfRec { hipEvent_t e0, e1, e2; int64_t samples; int path; };
  int profiling;
  std::vector<ProfRec> prof;
};

// which kernels a compute call ran (cnf_model_last_path)
enum CnfPath {
  CNF_PATH_NONE = 0,
  CNF_PATH_MLP1 = 1,        // flow_kernel, one sample per lane
  CNF_PATH_MLP2 = 2,        // flow_kernel, packed fp32, two samples per lane
  CNF_PATH_MFMA = 3,        // flow_kernel with the MFMA conditioner
  CNF_PATH_TABLES = 4,      // pwl_build_kernel + flow_pwl_kernel
  CNF_PATH_LOSS_MLP = 5,    // loss_kernel
  CNF_PATH_LOSS_TABLES = 6, // pwl_build_kernel + loss_pwl_kernel
  CNF_PATH_F64 = 7,         // float64 instantiation
  CNF_PATH_DPAR = 9,        // flow_dpar_kernel: one wave per conditioned dimension (dim >= 3, base -> data)
  CNF_PATH_DETECT = 8,      // per-sample condition: uniformity check + table kernels + MLP kernel, gated on the device
};

#ifdef CNF_MINIMAL_CONFIGS   /* faster builds while iterating on the kernels */
#define CNF_KERNEL_CONFIGS(X) X(16, 5)
#else
#define CNF_KERNEL_CONFIGS(X) \
  X(8, 5) X(16, 4) X(16, 5) X(16, 8) X(16, 10) X(32, 5) X(32, 8) X(64, 5)
#endif

// Dynamic LDS above the 64 KB default needs an explicit opt-in per kernel; the
// CU has 160 KB.  Returns false if the request cannot be met.
// The attribute is set once per (kernel, device) for the whole CU (a launch then needs no runtime call but
// the launch itself).
template <class K>
static inline bool ensure_lds(K kernel, size_t bytes) {
  if (bytes > 160 * 1024) return false;
  if (bytes <= 64 * 1024) return true;
  static std::mutex mu;
  // keyed by (kernel, device): kernels of one signature share this instantiation
  static std::map<std::pair<const void*, int>, bool> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_pair((const void*)kernel, dev);
  auto it = done.find(key);
  if (it != done.end()) return it->second;
  const bool ok = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
  done[key] = ok;
  return ok;
}

// Workgroups of `kernel` (threads, dynamic LDS bytes) that fit on one CU at a time (registers, LDS, waves), cached.
template <class K>
static inline int resident_blocks_per_cu(K kernel, int threads, size_t lds) {
  static std::mutex mu;
  static std::map<std::tuple<const void*, int, size_t, int>, int> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 1;
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_tuple((const void*)kernel, threads, lds, dev);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)kernel, threads, lds) != hipSuccess || n < 1) n = 1;
  // The API has been seen one workgroup per CU high (MI355X_MICROARCH.md: kernels with 81-112 SGPRs; here: 6 reported
  // for 27.8 KB of LDS per workgroup, 5 resident -- profiles/r03b_cfg4), and a grid one round larger than what is
  // resident costs a whole extra round: bound it by the two limits that can be computed here.
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, (const void*)kernel) == hipSuccess) {
    const size_t per_wg = ((lds + fa.sharedSizeBytes + 511) / 512) * 512;              // LDS allocation granularity
    if (per_wg > 0) { const int by_lds = (int)((160 * 1024) / per_wg); if (by_lds >= 1 && by_lds < n) n = by_lds; }
    const int regs = ((fa.numRegs > 0 ? fa.numRegs : 1) + 7) / 8 * 8;                  // unified VGPR file: 512 per SIMD lane
    int waves_per_simd = 512 / regs; if (waves_per_simd > 8) waves_per_simd = 8;
    const int by_regs = waves_per_simd * 4 / ((threads + 63) / 64);
    if (by_regs >= 1 && by_regs < n) n = by_regs;
  }
  cache[key] = n;
  return n;
}

// Grid of a kernel that walks `n_tiles` equal tiles with a grid-stride loop: as many workgroups as are resident at
// once (`capacity`), never more -- a grid of 2 048 single-wave workgroups on 1 536 slots ran a second, quarter-filled
// round as long as the first (profiles/r03a_cfg4).  With every slot taken the tiles that do not divide evenly (config
// 4: 10 923 tiles on 2 048 slots) end as a thinly populated last round in which a wave has its SIMD to itself and runs
// ~1.7 x as fast; giving every workgroup the same count instead (1 821 workgroups x 6 tiles) cost config 4's
// value_and_grad 7 % (1.56 vs 1.455 ms).
static inline int64_t balanced_grid(int64_t n_tiles, int64_t capacity) {
  if (capacity < 1) capacity = 1;
  if (n_tiles < 1) n_tiles = 1;
  return n_tiles < capacity ? n_tiles : capacity;
}

// Orders a compute call after the last cnf_model_set_params when that ran on a different stream.
static inline int wait_for_params(CnfModel* m, hipStream_t stream) {
  if (m->prep_event && m->prep_stream != (void*)stream)
    if (hipStreamWaitEvent(stream, m->prep_event, 0) != hipSuccess) return CNF_ERR_HIP;
  return CNF_OK;
}

// cnf_flow.hip: builds the dim-2 conditioner tables (cnf_pwl.h) of n slices -- conditions c[0 .. n) on the device --
// into the stream's reserved workspace and returns them; CNF_ERR_UNSUPPORTED if the configuration has no table path
// or the stream's reservation (cnf_model_reserve) is smaller than n.  Used by the table form of cnf_pass_vjp.
int cnf_internal_build_tables(CnfModel* m, hipStream_t stream, const float* c, int64_t n, float** tables);
int cnf_internal_flow_shared(CnfModel* m, hipStream_t stream, const float* in, const float* c, int64_t slice_len,
                             int64_t n_slices, const float* tables, float* out);

static inline cnf::ModelArgs model_args(const CnfModel* m) {
  cnf::ModelArgs a;
  a.prep = m->prep; a.per_layer = m->per_layer;
  a.wq = m->mfma_off > 0 ? m->prep + m->mfma_off : nullptr; a.per_layer_q = m->per_layer_q;
  a.D = m->cfg.dim; a.L = m->cfg.num_layers; a.M = m->cfg.mlp_num_layers;
  a.sc = m->sc; a.scd = m->scd;
  a.tabd = reinterpret_cast<const double*>(m->prep + m->tabd_off);
  a.e2tab = reinterpret_cast<const double*>(m->prep + m->e2_off);
  return a;
}
