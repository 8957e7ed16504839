// Measurement aid (not product code): what an MFMA costs by the register class of its operands, one wave per SIMD.
//   C / D in AGPRs or VGPRs  x  B operand in a VGPR or an AGPR  x  0 / 2 / 4 independent v_fma_f32 after every MFMA
// Question behind it (round 4): the whole-layer kernel moves every chunk accumulator through v_accvgpr_read (2 issue
// slots each) before the GeGLU; would VGPR accumulators (no moves) cost more at the MFMA than the moves do?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o acc_class_probe acc_class_probe.hip && ./acc_class_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                     \
  do {                                                               \
    hipError_t e_ = (x);                                             \
    if (e_ != hipSuccess) {                                          \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); \
      exit(1);                                                       \
    }                                                                \
  } while (0)

// ACC: 0 = "+a", 1 = "+v";  BCLS: 0 = b in a VGPR, 1 = b in an AGPR;  NV: v_fma_f32 per MFMA;  CHAIN: accumulators in
// rotation (8 = independent, 4 = as a chunk step of the layer kernel: an accumulator every fourth MFMA)
template <int ACC, int BCLS, int NV, int CHAIN>
__global__ __launch_bounds__(256) void probe_kernel(int n_iters, unsigned long long* __restrict__ out, float* __restrict__ sink) {
  __shared__ unsigned short pad[48 * 1024];  // 96 KiB: one block per CU
  const int lane = threadIdx.x & 63;
  pad[threadIdx.x] = (unsigned short)lane;
  f16x8 av[4], bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      av[j][i] = (_Float16)(0.001f * (float)((lane * 7 + i * 3 + j) % 61));
      bv[j][i] = (_Float16)(0.002f * (float)(((lane ^ i) * 5 + j) % 53));
    }
    asm volatile("" : "+v"(av[j]));
    if (BCLS) asm volatile("" : "+a"(bv[j]));
    else asm volatile("" : "+v"(bv[j]));
  }
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float fv[8], fc = 1.0001f;
#pragma unroll
  for (int i = 0; i < 8; ++i) fv[i] = 0.5f + (float)lane * 0.001f + (float)i;
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long t0 = wall_clock64();
  for (int it = 0; it < n_iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f32x4& c = acc[i % CHAIN];
        const f16x8 a = av[i & 3];
        if (ACC == 0 && BCLS == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(bv[(i + r) & 3]));
        if (ACC == 0 && BCLS == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "a"(bv[(i + r) & 3]));
        if (ACC == 1 && BCLS == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(bv[(i + r) & 3]));
        if (ACC == 1 && BCLS == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(bv[(i + r) & 3]));
#pragma unroll
        for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fv[v & 7]) : "v"(fc));
      }
  }
  float total = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) total += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + fv[i];
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long t1 = wall_clock64();
  if (total == 123.456f) sink[threadIdx.x] = total + (float)pad[(threadIdx.x * 7) & 1023];
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = t1 - t0;
  }
}

template <int ACC, int BCLS, int NV, int CHAIN>
static void probe(const char* label, float* sink) {
  const int blocks = 256 * 4, n_iters = 2048;
  unsigned long long* out;
  CHECK(hipMalloc(&out, (size_t)blocks * 2 * sizeof(unsigned long long)));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe_kernel<ACC, BCLS, NV, CHAIN>), dim3(blocks), dim3(256), 0, 0, n_iters, out, sink);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned long long> host((size_t)blocks * 2);
  CHECK(hipMemcpy(host.data(), out, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double cyc = 0.0, real = 0.0;
  for (int b = 0; b < blocks; ++b) {
    cyc += (double)host[2 * b];
    real += (double)host[2 * b + 1];
  }
  printf("%-64s %6.3f GHz, %6.2f cycles per MFMA\n", label, cyc / real * 0.1, cyc / blocks / ((double)n_iters * 32));
  CHECK(hipFree(out));
}

int main() {
  float* sink;
  CHECK(hipMalloc(&sink, 4096));
  printf("v_mfma_f32_16x16x32_f16, one wave per SIMD, by operand register class (C/D, B), fillers and accumulator rotation\n");
  probe<0, 0, 0, 8>("C/D AGPR, B VGPR, 8 accumulators, 0 v_fma", sink);
  probe<0, 1, 0, 8>("C/D AGPR, B AGPR, 8 accumulators, 0 v_fma", sink);
  probe<1, 0, 0, 8>("C/D VGPR, B VGPR, 8 accumulators, 0 v_fma", sink);
  probe<1, 1, 0, 8>("C/D VGPR, B AGPR, 8 accumulators, 0 v_fma", sink);
  probe<0, 0, 2, 8>("C/D AGPR, B VGPR, 8 accumulators, 2 v_fma", sink);
  probe<0, 1, 2, 8>("C/D AGPR, B AGPR, 8 accumulators, 2 v_fma", sink);
  probe<1, 0, 2, 8>("C/D VGPR, B VGPR, 8 accumulators, 2 v_fma", sink);
  probe<1, 1, 2, 8>("C/D VGPR, B AGPR, 8 accumulators, 2 v_fma", sink);
  probe<0, 0, 4, 8>("C/D AGPR, B VGPR, 8 accumulators, 4 v_fma", sink);
  probe<0, 1, 4, 8>("C/D AGPR, B AGPR, 8 accumulators, 4 v_fma", sink);
  probe<1, 0, 4, 8>("C/D VGPR, B VGPR, 8 accumulators, 4 v_fma", sink);
  probe<1, 1, 4, 8>("C/D VGPR, B AGPR, 8 accumulators, 4 v_fma", sink);
  probe<0, 0, 0, 4>("C/D AGPR, B VGPR, 4 accumulators, 0 v_fma", sink);
  probe<1, 0, 0, 4>("C/D VGPR, B VGPR, 4 accumulators, 0 v_fma", sink);
  probe<1, 1, 0, 4>("C/D VGPR, B AGPR, 4 accumulators, 0 v_fma", sink);
  probe<0, 0, 2, 4>("C/D AGPR, B VGPR, 4 accumulators, 2 v_fma", sink);
  probe<1, 0, 2, 4>("C/D VGPR, B VGPR, 4 accumulators, 2 v_fma", sink);
  probe<1, 1, 2, 4>("C/D VGPR, B AGPR, 4 accumulators, 2 v_fma", sink);
  return 0;
}
