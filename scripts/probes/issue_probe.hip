// issue_probe.hip -- issue cost of the vector instructions the flow kernels are made of, on one SIMD.
// Each kernel runs `iters` iterations of 8 INDEPENDENT chains of one instruction (inline asm), with W waves
// per SIMD; cycles per instruction = elapsed * clock / (iters * 8 * W).  Printed relative to v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define CHAIN8(STMT) STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7)

#define PROBE_SCALAR(NAME, ASM)                                                        \
  __global__ void NAME(float* out, int iters) {                                        \
    float a[8];                                                                        \
    for (int i = 0; i < 8; ++i) a[i] = 1.0f + 1e-3f * (threadIdx.x + i);               \
    const float b = 1.0000001f, c = 1e-9f;                                             \
    for (int it = 0; it < iters; ++it) {                                               \
      CHAIN8(ASM)                                                                      \
    }                                                                                  \
    float s = 0.f;                                                                     \
    for (int i = 0; i < 8; ++i) s += a[i];                                             \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                    \
  }

#define PROBE_PACKED(NAME, ASM)                                                        \
  __global__ void NAME(float* out, int iters) {                                        \
    v2f a[8];                                                                          \
    for (int i = 0; i < 8; ++i) a[i] = v2f{1.0f + 1e-3f * (threadIdx.x + i), 1.5f};    \
    const v2f b = {1.0000001f, 0.9999999f}, c = {1e-9f, 1e-9f};                        \
    for (int it = 0; it < iters; ++it) {                                               \
      CHAIN8(ASM)                                                                      \
    }                                                                                  \
    float s = 0.f;                                                                     \
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;                                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                    \
  }

#define S_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define S_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define S_LOG(i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
#define S_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define S_SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define S_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
#define S_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
#define S_CND64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b) : "s20", "s21");
#define S_CMP(i) asm volatile("v_cmp_ge_f32_e64 s[22:23], %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c) : "s22", "s23");
#define S_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define S_CMPCND(i) asm volatile("v_cmp_ge_f32_e64 s[22:23], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(a[i]) : "v"(b) : "s22", "s23");
#define P_FMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define P_MUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define P_ADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define P_MULC(i) asm volatile("v_pk_mul_f32 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(b));

#define PROBE_DOUBLE(NAME, ASM)                                                        \
  __global__ void NAME(float* out, int iters) {                                        \
    double a[8];                                                                       \
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + 1e-3 * (threadIdx.x + i);                 \
    const double b = 1.0000001, c = 1e-9;                                              \
    for (int it = 0; it < iters; ++it) {                                               \
      CHAIN8(ASM)                                                                      \
    }                                                                                  \
    double s = 0.;                                                                     \
    for (int i = 0; i < 8; ++i) s += a[i];                                             \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;                             \
  }
#define D_FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define D_ADD(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define D_MUL(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define D_RCP(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
// conversions: a double chain element is narrowed and widened again (two instructions per step)
#define D_CVT(i) { float t_; asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(t_) : "v"(a[i])); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(t_)); }
#define S_CVTU(i) { unsigned t_; asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(t_) : "v"(a[i])); asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(t_)); }
#define S_RNDNE(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i]));
#define S_LDEXP(i) asm volatile("v_ldexp_f32 %0, %0, 1" : "+v"(a[i]));
#define S_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(c));
#define S_FLOOR(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
#define S_MIN(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define S_PKMOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
PROBE_DOUBLE(k_dfma, D_FMA)
PROBE_DOUBLE(k_dadd, D_ADD)
PROBE_DOUBLE(k_dmul, D_MUL)
PROBE_DOUBLE(k_drcp, D_RCP)
PROBE_DOUBLE(k_dcvt, D_CVT)
PROBE_SCALAR(k_cvtu, S_CVTU)
PROBE_SCALAR(k_rndne, S_RNDNE)
PROBE_SCALAR(k_ldexp, S_LDEXP)
PROBE_SCALAR(k_lshladd, S_LSHLADD)
PROBE_SCALAR(k_floor, S_FLOOR)
PROBE_SCALAR(k_min, S_MIN)
PROBE_SCALAR(k_mov, S_PKMOV)
PROBE_SCALAR(k_fma, S_FMA)
PROBE_SCALAR(k_mul, S_MUL)
PROBE_SCALAR(k_exp, S_EXP)
PROBE_SCALAR(k_log, S_LOG)
PROBE_SCALAR(k_rcp, S_RCP)
PROBE_SCALAR(k_sqrt, S_SQRT)
PROBE_SCALAR(k_max3, S_MAX3)
PROBE_SCALAR(k_med3, S_MED3)
PROBE_SCALAR(k_cnd, S_CND)
PROBE_SCALAR(k_add, S_ADD)
PROBE_SCALAR(k_cmp_add, S_CMP)
PROBE_SCALAR(k_cmp_cnd, S_CMPCND)
__global__ void k_cnd64(float* out, int iters) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = 1.0f + 1e-3f * (threadIdx.x + i);
  const float b = 1.0000001f;
  asm volatile("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x55555555" ::: "s20", "s21");
  for (int it = 0; it < iters; ++it) {
    CHAIN8(S_CND64)
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
PROBE_PACKED(k_pk_fma, P_FMA)
PROBE_PACKED(k_pk_mul, P_MUL)
PROBE_PACKED(k_pk_add, P_ADD)
PROBE_PACKED(k_pk_mul_clamp, P_MULC)

typedef void (*kern_t)(float*, int);

static double run(kern_t k, int waves_per_simd, int iters, float* out, int num_cus) {
  const int threads = 64 * 4 * waves_per_simd;        // 4 SIMDs per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(num_cus), dim3(threads), 0, 0, out, 16);      // warm-up
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(num_cus), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int num_cus = prop.multiProcessorCount;
  const double clock_hz = prop.clockRate * 1e3;        // kHz -> Hz (the nominal peak clock)
  float* out;
  hipMalloc(&out, sizeof(float) * (size_t)num_cus * 1024);
  const int iters = 200000;
  struct { const char* name; kern_t k; } ks[] = {
    {"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_max3_f32", k_max3}, {"v_med3_f32", k_med3}, {"v_cndmask_b32 (vcc)", k_cnd},
    {"v_cndmask_b32_e64 (sgpr mask)", k_cnd64}, {"v_add_f32", k_add}, {"v_cmp_ge_f32_e64 + v_add_f32", k_cmp_add},
    {"v_cmp_ge_f32_e64 + v_cndmask_e64", k_cmp_cnd},
    {"v_pk_fma_f32", k_pk_fma}, {"v_pk_mul_f32", k_pk_mul}, {"v_pk_add_f32", k_pk_add}, {"v_pk_mul_f32 clamp", k_pk_mul_clamp},
    {"v_exp_f32", k_exp}, {"v_log_f32", k_log}, {"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt},
    {"v_fma_f64", k_dfma}, {"v_add_f64", k_dadd}, {"v_mul_f64", k_dmul}, {"v_rcp_f64", k_drcp},
    {"v_cvt_f32_f64 + v_cvt_f64_f32", k_dcvt}, {"v_cvt_u32_f32 + v_cvt_f32_u32", k_cvtu},
    {"v_rndne_f32", k_rndne}, {"v_floor_f32", k_floor}, {"v_ldexp_f32", k_ldexp}, {"v_lshl_add_u32", k_lshladd},
    {"v_min_f32", k_min}, {"v_mov_b32", k_mov},
  };
  printf("%d CUs, nominal clock %.0f MHz; cycles per wave64 instruction per SIMD (at the nominal clock)\n", num_cus, clock_hz / 1e6);
  for (int w : {1, 2, 4}) {
    double ref = 0;
    for (auto& e : ks) {
      const double t = run(e.k, w, iters, out, num_cus);
      const double cyc = t * clock_hz / ((double)iters * 8 * w);
      if (e.k == k_fma) ref = cyc;
      printf("  %d wave(s)/SIMD  %-34s %6.2f cycles  (%.2f x v_fma_f32)\n", w, e.name, cyc, cyc / ref);
    }
  }
  hipFree(out);
  return 0;
}
