// Check of the operand / result layout of v_mfma_f32_32x32x16_bf16 assumed by opk_layer32.hip.h:
//   A (32 x 16): lane l holds row l % 32, k = 8 (l / 32) + (0..7);  B (16 x 32): column l % 32, same k;
//   D (32 x 32): lane l holds column l % 32, rows 8 (i / 4) + 4 (l / 32) + i % 4 in register i.
//   hipcc --offload-arch=gfx950 -O2 -o mfma32_layout mfma32_layout.hip && ./mfma32_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* D) {
  const int l = threadIdx.x, n = l & 31, h = l >> 5;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (__bf16)A[n * 16 + 8 * h + e];
    b[e] = (__bf16)B[(8 * h + e) * 32 + n];
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) D[(8 * (i / 4) + 4 * h + i % 4) * 32 + n] = c[i];
}
int main() {
  float hA[32 * 16], hB[16 * 32], hD[32 * 32], ref[32 * 32];
  for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7 + 3) % 13 - 6); hB[i] = (float)((i * 5 + 1) % 11 - 5); }
  for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += hA[m * 16 + kk] * hB[kk * 32 + n]; ref[m * 32 + n] = s; }
  float *dA, *dB, *dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 1024; ++i) bad += hD[i] != ref[i];
  printf("mfma 32x32x16 layout: %s (%d of 1024 elements differ)\n", bad ? "MISMATCH" : "as assumed", bad);
  return bad != 0;
}
