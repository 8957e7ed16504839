// mode_probe.hip -- MODE.FP16_OVFL on gfx950: (1) v_cvt_pk_f16_f32 of 1e5 / NaN / inf under each setting, switching back and forth inside a
// kernel; (2) v_mfma_f32_16x16x32_f16 with a NaN / inf operand element under each setting.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pack(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__global__ void probe(const float* in, unsigned* out, float* mf) {
  float a = in[threadIdx.x], b = in[threadIdx.x] * 2.f;
  out[0 * 64 + threadIdx.x] = pack(a, b);  // default mode
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  asm volatile("" : "+v"(a), "+v"(b));
  out[1 * 64 + threadIdx.x] = pack(a, b);  // saturating
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0");
  asm volatile("" : "+v"(a), "+v"(b));
  out[2 * 64 + threadIdx.x] = pack(a, b);  // overflowing again
  // MFMA: A = ones except element 0 of lane 0 = in[64] (NaN) / in[65] (inf); B = ones
  for (int which = 0; which < 4; ++which)  // 0 / 1: NaN / inf in the FIRST operand, 2 / 3: in the second
    for (int mode = 0; mode < 2; ++mode) {
      f16x8 x, y;
      for (int i = 0; i < 8; ++i) { x[i] = (_Float16)1.0f; y[i] = (_Float16)1.0f; }
      if (threadIdx.x == 0 && which < 2) x[0] = (_Float16)in[64 + which];
      if (threadIdx.x == 0 && which >= 2) y[0] = (_Float16)in[64 + which - 2];
      if (mode) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
      else asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0");
      asm volatile("" : "+v"(x), "+v"(y));
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c, 0, 0, 0);
      asm volatile("" : "+v"(c));
      mf[((which * 2 + mode) * 64 + threadIdx.x) * 4 + 0] = c[0];
      mf[((which * 2 + mode) * 64 + threadIdx.x) * 4 + 1] = c[1];
      mf[((which * 2 + mode) * 64 + threadIdx.x) * 4 + 2] = c[2];
      mf[((which * 2 + mode) * 64 + threadIdx.x) * 4 + 3] = c[3];
    }
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0");
}
int main() {
  float h[66];
  for (int i = 0; i < 64; ++i) h[i] = 1.0e5f;
  h[1] = __builtin_nanf(""); h[2] = __builtin_inff();
  h[64] = __builtin_nanf(""); h[65] = __builtin_inff();
  float* d_in; unsigned* d_out; float* d_mf;
  hipMalloc(&d_in, sizeof(h)); hipMalloc(&d_out, 3 * 64 * 4); hipMalloc(&d_mf, 8 * 64 * 4 * 4);
  hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_in, d_out, d_mf);
  unsigned o[192]; float m[8 * 64 * 4];
  hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
  hipMemcpy(m, d_mf, sizeof(m), hipMemcpyDeviceToHost);
  printf("cvt 1e5: default %08x  FP16_OVFL=1 %08x  back to 0 %08x   (7c00 = inf, 7bff = 65504)\n", o[0], o[64], o[128]);
  printf("cvt NaN: %08x %08x %08x | inf: %08x %08x %08x\n", o[1], o[65], o[129], o[2], o[66], o[130]);
  const char* names[8] = {"NaN in operand 1, FP16_OVFL=0", "NaN in operand 1, FP16_OVFL=1", "inf in operand 1, FP16_OVFL=0", "inf in operand 1, FP16_OVFL=1",
                          "NaN in operand 2, FP16_OVFL=0", "NaN in operand 2, FP16_OVFL=1", "inf in operand 2, FP16_OVFL=0", "inf in operand 2, FP16_OVFL=1"};
  for (int k = 0; k < 8; ++k) {
    int n_nan = 0, n_inf = 0; float sample = 0.f;
    for (int i = 0; i < 256; ++i) { float v = m[k * 256 + i]; if (v != v) ++n_nan; else if (v > 1e30f || v < -1e30f) ++n_inf; else if (v != 32.f) sample = v; }
    printf("mfma 16x16x32 f16, %s: %d NaN, %d inf of 256 outputs (a finite output other than 32: %g)\n", names[k], n_nan, n_inf, sample);
  }
  return 0;
}
