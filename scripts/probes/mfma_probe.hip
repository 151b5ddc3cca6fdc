// Probe: v_mfma_f32_16x16x4_f32 operand/result lane maps and the gfx950
// v_permlane{16,32}_swap semantics that conditioner_mfma relies on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* A /*16x4 row-major [i][k]*/, const float* B /*4x16 [k][n]*/, float* D /*16x16 [i][n]*/,
                      unsigned* sw /*4 x 64*/) {
  const int l = threadIdx.x;
  const float a = A[(l & 15) * 4 + (l >> 4)];     // lane l: A[i=l&15][k=l>>4]
  const float b = B[(l >> 4) * 16 + (l & 15)];    // lane l: B[k=l>>4][n=l&15]
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];   // row=(l>>4)*4+r, col=l&15
  unsigned x = 100 + l, y = 200 + l;
  auto r32 = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  sw[0 * 64 + l] = r32[0]; sw[1 * 64 + l] = r32[1];
  auto r16 = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  sw[2 * 64 + l] = r16[0]; sw[3 * 64 + l] = r16[1];
}

int main() {
  float hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 64; ++i) { hA[i] = (float)((i * 7) % 11) - 3.f; hB[i] = (float)((i * 5) % 13) - 6.f; }
  for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) { float s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + n]; ref[i * 16 + n] = s; }
  float *dA, *dB, *dD; unsigned* dS; unsigned hS[256];
  hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 1024); hipMalloc(&dS, 1024);
  hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dD, dS);
  hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost); hipMemcpy(hS, dS, 1024, hipMemcpyDeviceToHost);
  double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(hD[i] - ref[i]));
  printf("mfma max err %g\n", err);
  const char* names[4] = {"p32.x", "p32.y", "p16.x", "p16.y"};
  for (int r = 0; r < 4; ++r) { printf("%s:", names[r]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, hS[r * 64 + l]); printf("\n"); }
  return err < 1e-5 ? 0 : 1;
}
