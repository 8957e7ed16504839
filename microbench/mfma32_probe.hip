// Microbenchmark (measurement aid, not product code): how much non-MFMA issue fits beside one wave's MFMA stream for
// the two bf16 shapes of gfx950 -- v_mfma_f32_16x16x32_bf16 (4 passes, 16 cycles) and v_mfma_f32_32x32x16_bf16
// (8 passes, 32 cycles, the same flop rate) -- with the accumulators in AGPRs, one wave per SIMD, and a fixed mix of
// vector / accumulator-read / LDS instructions per unit of MFMA work (one unit = 32768 flop: two 16x16x32 or one
// 32x32x16).  Also: vector-only throughput of scalar vs packed fp32 arithmetic (the LayerNorm phases).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma32_probe mfma32_probe.hip && ./mfma32_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

#define CHECK(x)                                                     \
  do {                                                               \
    hipError_t e_ = (x);                                             \
    if (e_ != hipSuccess) {                                          \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); \
      exit(1);                                                       \
    }                                                                \
  } while (0)

// SHAPE 0: 16x16x32 (two per unit), 1: 32x32x16 (one per unit), 2: no MFMA at all.
// Per unit: NF independent v_fma_f32, NA v_accvgpr_read_b32, NC v_cvt_pk_bf16_f32, NL ds_read_b128 (x1/2: every other
// unit when NL == -1), NP v_pk_fma_f32, NX v_exp_f32, ND dependent v_fma (one chain).
template <int SHAPE, int NF, int NA, int NC, int NL, int NP, int NX, int ND>
__global__ __launch_bounds__(256) void probe_kernel(int n_iters, unsigned long long* __restrict__ out, float* __restrict__ sink) {
  __shared__ u16 pad[48 * 1024];  // 96 KiB: one block per CU
  const int lane = threadIdx.x & 63;
  pad[threadIdx.x] = (u16)lane;
  bf16x8 av[4], bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      av[j][i] = (__bf16)(0.001f * (float)(lane + i + j));
      bv[j][i] = (__bf16)(0.002f * (float)((lane ^ i) + j));
    }
    asm volatile("" : "+v"(av[j]), "+v"(bv[j]));
  }
  f32x4 acc16[8];
  f32x16 acc32[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc16[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
  float fv[8], fc = 1.0001f, spare[4] = {1.f, 2.f, 3.f, 4.f};
  f32x2 pv[8], pc = f32x2{1.0001f, 0.9999f};
  unsigned pk[4] = {0, 0, 0, 0};
  uint4 lv[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
  const unsigned lds_a = (unsigned)(uintptr_t)((__attribute__((address_space(3))) u16*)pad) + lane * 16u;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    fv[i] = 0.5f + (float)lane * 0.001f + (float)i;
    pv[i] = f32x2{0.5f + (float)lane, 0.25f + (float)i};
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(spare[i]));
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < n_iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {  // 16 units per iteration
      auto extras = [&](int half) {
        // the extras of a unit are split evenly behind its MFMAs (two halves for the 16x16 shape)
        const int parts = SHAPE == 0 ? 2 : 1;
#pragma unroll
        for (int v = 0; v < NF; ++v)
          if (v % parts == half) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fv[v & 7]) : "v"(fc));
#pragma unroll
        for (int v = 0; v < NA; ++v)
          if (v % parts == half) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(fv[(v + 3) & 7]) : "a"(spare[v & 3]));
#pragma unroll
        for (int v = 0; v < NC; ++v)
          if (v % parts == half) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[v & 3]) : "v"(fv[v & 7]), "v"(fc));
        if (NL > 0) {
#pragma unroll
          for (int v = 0; v < NL; ++v)
            if (v % parts == half) asm volatile("ds_read_b128 %0, %1" : "=v"(lv[v & 1]) : "v"(lds_a));
        } else if (NL == -1) {
          if ((u & 1) == 0 && half == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(lv[0]) : "v"(lds_a));
        }
#pragma unroll
        for (int v = 0; v < NP; ++v)
          if (v % parts == half) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pv[v & 7]) : "v"(pc));
#pragma unroll
        for (int v = 0; v < NX; ++v)
          if (v % parts == half) asm volatile("v_exp_f32 %0, %0" : "+v"(fv[(v + 5) & 7]));
#pragma unroll
        for (int v = 0; v < ND; ++v)
          if (v % parts == half) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fv[0]) : "v"(fc));
      };
      if (SHAPE == 0) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc16[(2 * u) & 7]) : "v"(av[u & 3]), "v"(bv[(u >> 2) & 3]));
        extras(0);
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc16[(2 * u + 1) & 7]) : "v"(av[u & 3]), "v"(bv[(u >> 2) & 3]));
        extras(1);
      } else if (SHAPE == 1) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc32[u & 3]) : "v"(av[u & 3]), "v"(bv[(u >> 2) & 3]));
        extras(0);
      } else {
        extras(0);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float total = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) total += acc16[i][0] + acc16[i][3] + fv[i] + pv[i][0] + pv[i][1];
#pragma unroll
  for (int i = 0; i < 4; ++i) total += acc32[i][0] + acc32[i][15];
  total += (float)(pk[0] ^ pk[1] ^ pk[2] ^ pk[3]) + (float)(lv[0].x ^ lv[1].y);
  const unsigned long long c1 = __builtin_readcyclecounter();
  if (total == 123.456f) sink[threadIdx.x] = total + (float)pad[(threadIdx.x * 7) & 1023];
  if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
}

template <int SHAPE, int NF, int NA = 0, int NC = 0, int NL = 0, int NP = 0, int NX = 0, int ND = 0>
static void probe(const char* label, float* sink) {
  const int blocks = 256 * 4, n_iters = 2048;
  unsigned long long* out;
  CHECK(hipMalloc(&out, (size_t)blocks * sizeof(unsigned long long)));
  for (int i = 0; i < 2; ++i)
    hipLaunchKernelGGL((probe_kernel<SHAPE, NF, NA, NC, NL, NP, NX, ND>), dim3(blocks), dim3(256), 0, 0, n_iters, out, sink);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned long long> host((size_t)blocks);
  CHECK(hipMemcpy(host.data(), out, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double cyc = 0.0;
  for (int b = 0; b < blocks; ++b) cyc += (double)host[b];
  const double per_unit = cyc / blocks / ((double)n_iters * 16);
  const char* shape = SHAPE == 0 ? "2 x 16x16x32" : (SHAPE == 1 ? "1 x 32x32x16" : "no MFMA     ");
  printf("%s  %-52s %7.2f cycles per unit%s\n", shape, label, per_unit, SHAPE < 2 ? (per_unit < 33.5 ? "  (MFMA-bound)" : "") : "");
  CHECK(hipFree(out));
}


// Power probe: the MFMA stream alone, every SIMD of the chip busy (one or two waves per SIMD), operands either a few
// constants (as above) or pseudo-random bf16 values, 16 distinct (a, b) register pairs in rotation.  Reports the shader
// clock the chip holds (s_memtime / s_memrealtime) and the executed TFLOP/s: with random operands the matrix pipes
// toggle and the power limit, not the issue rate, sets the throughput.
template <int SHAPE, int WPS>
__global__ __launch_bounds__(256 * WPS) void power_kernel(int n_iters, int random, unsigned long long* __restrict__ out, float* __restrict__ sink) {
  __shared__ u16 pad[(WPS == 1 ? 48 : 16) * 1024];
  const int lane = threadIdx.x & 63;
  pad[threadIdx.x] = (u16)lane;
  bf16x8 av[8], bv[8];
  unsigned s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s = s * 1664525u + 1013904223u;
      const float fa = random ? ((int)(s >> 9) / 8388608.0f - 0.5f) * 2.0f : 0.001f * (float)(lane + i + j);
      s = s * 1664525u + 1013904223u;
      const float fb = random ? ((int)(s >> 9) / 8388608.0f - 0.5f) * 0.125f : 0.002f * (float)((lane ^ i) + j);
      av[j][i] = (__bf16)fa;
      bv[j][i] = (__bf16)fb;
    }
    asm volatile("" : "+v"(av[j]), "+v"(bv[j]));
  }
  f32x4 acc16[8];
  f32x16 acc32[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc16[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long t0 = wall_clock64();
  for (int it = 0; it < n_iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (SHAPE == 0) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc16[(2 * u) & 7]) : "v"(av[u & 7]), "v"(bv[(u >> 1) & 7]));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc16[(2 * u + 1) & 7]) : "v"(av[(u + 3) & 7]), "v"(bv[(u >> 1) & 7]));
      } else {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc32[u & 3]) : "v"(av[u & 7]), "v"(bv[(u >> 1) & 7]));
      }
    }
  }
  float total = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) total += acc16[i][0] + acc16[i][3];
#pragma unroll
  for (int i = 0; i < 4; ++i) total += acc32[i][0] + acc32[i][15];
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long t1 = wall_clock64();
  if (total == 123.456f) sink[threadIdx.x] = total + (float)pad[(threadIdx.x * 7) & 1023];
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = t1 - t0;
  }
}

template <int SHAPE, int WPS>
static void power_probe(const char* label, int random, float* sink) {
  const int blocks = 256 * 8, n_iters = 8192;
  unsigned long long* out;
  CHECK(hipMalloc(&out, (size_t)blocks * 2 * sizeof(unsigned long long)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((power_kernel<SHAPE, WPS>), dim3(blocks), dim3(256 * WPS), 0, 0, n_iters, random, out, sink);
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((power_kernel<SHAPE, WPS>), dim3(blocks), dim3(256 * WPS), 0, 0, n_iters, random, out, sink);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> host((size_t)blocks * 2);
  CHECK(hipMemcpy(host.data(), out, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double cyc = 0.0, real = 0.0;
  for (int b = 0; b < blocks; ++b) {
    cyc += (double)host[2 * b];
    real += (double)host[2 * b + 1];
  }
  const double flops = (double)blocks * 4 * WPS * (double)n_iters * 16 * 32768.0;  // 16 units of 32768 flop per iteration
  printf("%s  %d wave(s)/SIMD  %-18s %6.3f GHz shader clock  %7.1f TFLOP/s executed  (%.1f ms)\n",
         SHAPE == 0 ? "16x16x32" : "32x32x16", WPS, label, cyc / real * 0.1, flops / ms * 1e-9, ms);
  CHECK(hipFree(out));
}

#define BOTH(label, ...)               \
  probe<0, __VA_ARGS__>(label, sink);  \
  probe<1, __VA_ARGS__>(label, sink)

int main(int argc, char** argv) {
  float* sink;
  CHECK(hipMalloc(&sink, 4096));
  if (argc > 1) {  // "power": the data-dependent power limit of the matrix pipes
    printf("MFMA stream alone on every SIMD, ~0.2 s per case:\n");
    for (int random = 0; random < 2; ++random) {
      const char* label = random ? "random operands" : "constant operands";
      power_probe<0, 1>(label, random, sink);
      power_probe<1, 1>(label, random, sink);
      power_probe<0, 2>(label, random, sink);
      power_probe<1, 2>(label, random, sink);
    }
    return 0;
  }
  printf("one wave per SIMD, AGPR accumulators; unit = 32768 flop of MFMA work (32 pipe cycles)\n");
  BOTH("nothing else", 0);
  BOTH("2 v_fma", 2);
  BOTH("4 v_fma", 4);
  BOTH("6 v_fma", 6);
  BOTH("8 v_fma", 8);
  BOTH("10 v_fma", 10);
  BOTH("12 v_fma", 12);
  BOTH("2 v_accvgpr_read", 0, 2);
  BOTH("4 v_accvgpr_read", 0, 4);
  BOTH("2 v_cvt_pk_bf16_f32", 0, 0, 2);
  BOTH("4 v_cvt_pk_bf16_f32", 0, 0, 4);
  BOTH("1 ds_read_b128", 0, 0, 0, 1);
  BOTH("2 ds_read_b128", 0, 0, 0, 2);
  BOTH("2 v_exp_f32", 0, 0, 0, 0, 0, 2);
  BOTH("2 v_pk_fma_f32", 0, 0, 0, 0, 2);
  BOTH("4 v_pk_fma_f32", 0, 0, 0, 0, 4);
  BOTH("4 dependent v_fma", 0, 0, 0, 0, 0, 0, 4);
  // the fused MLP's macro-iteration per unit (192 16x16 MFMAs = 96 units: 206 vector, 56 accvgpr, 48 LDS reads)
  BOTH("MLP mix: 2 fma + 0.5 lds", 2, 0, 0, -1);
  BOTH("MLP mix: 2 fma + 1 acc-read + 0.5 lds", 2, 1, 0, -1);
  BOTH("MLP mix: 3 fma + 1 acc-read + 1 cvt + 0.5 lds", 3, 1, 1, -1);
  BOTH("MLP mix: 4 fma + 1 acc-read + 1 cvt + 0.5 lds", 4, 1, 1, -1);
  BOTH("MLP mix: 4 fma + 2 acc-read + 1 cvt + 1 exp + 0.5 lds", 4, 2, 1, -1, 0, 1);
  printf("vector-only (LayerNorm phases), one wave per SIMD:\n");
  probe<2, 8>("8 independent v_fma", sink);
  probe<2, 0, 0, 0, 0, 8>("8 independent v_pk_fma_f32 (16 fp32 FMAs)", sink);
  probe<2, 0, 0, 0, 0, 0, 0, 8>("8 dependent v_fma", sink);
  probe<2, 0, 8>("8 v_accvgpr_read", sink);
  probe<2, 0, 0, 8>("8 v_cvt_pk_bf16_f32", sink);
  probe<2, 4, 0, 4>("4 v_fma + 4 v_cvt_pk", sink);
  return 0;
}
