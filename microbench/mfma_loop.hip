// Microbenchmark (measurement aid, not product code): the steady-state loop of the row-stationary GEMM
// -- A operand in registers (32 rows x K=256, hi/lo planes), one 32-feature weight chunk (32 KiB) per iteration
// streamed global -> LDS by DMA into a two-stage ring, one block barrier per chunk -- built twice:
//   variant 0: v_mfma_f32_16x16x32_bf16  (96 MFMAs per chunk per wave, as open_provence_amd's rowgemm_kernel)
//   variant 1: v_mfma_f32_32x32x16_bf16  (48 MFMAs per chunk per wave)
// with a configurable amount of dummy epilogue VALU work per chunk (a Horner chain per accumulator value) and
// switches for the DMA and the barrier.  Prints ms and the bf16 TFLOP/s of the MFMA stream for each case.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_loop mfma_loop.hip && ./mfma_loop
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));               \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

constexpr int KS = 8;                  // K = 256
constexpr int STAGE = KS * 2 * 1024;   // u16 elements per stage: [ks][plane][2 fragments][512]

__device__ __forceinline__ bf16x8 lds_frag(const u16* p) { return *reinterpret_cast<const bf16x8*>(p); }

// Clock probe: N back-to-back independent MFMAs per wave (8 accumulators, operands in registers), timed with the
// shader clock (s_memtime, counts at the current core clock) and the constant 100 MHz counter (s_memrealtime).
//   cycles / MFMA / SIMD      -> issue rate (16 for v_mfma_f32_16x16x32_bf16 at one wave per SIMD)
//   shader cycles / real time -> the clock the chip actually runs at under this MFMA load
// The guide's 2.5 PFLOP/s dense bf16 peak is 1024 SIMDs x 1024 flop/cycle x 2.4 GHz; at the clock measured here the
// ceiling of this part is 1024 x 1024 x f, which is what the TFLOP/s figures printed below should be read against.
// MODE 0: one (a, b) register pair, accumulators wherever the compiler puts them; 1: accumulators forced into AGPRs
// (inline-asm MFMA with "+a"); 2: eight distinct (a, b) pairs; 3: both.
// NV: independent vector instructions issued after every MFMA (KIND 0: v_fma_f32, 1: v_accvgpr_read_b32 of a finished
// accumulator, 2: ds_read_b128 + nothing waiting on it, 3: v_cvt_pk_bf16_f32) -- what one wave can issue in an MFMA's shadow.
template <int WAVES_PER_SIMD, int MODE = 0, int NV = 0, int KIND = 0>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD > 1024 ? 1024 : 256 * WAVES_PER_SIMD) void clock_probe_kernel(
    int n_iters, unsigned long long* __restrict__ out, float* __restrict__ sink) {
  __shared__ u16 pad[48 * 1024];  // 96 KiB: one block per CU, so the block size sets the waves per SIMD
  const int lane = threadIdx.x & 63;
  pad[threadIdx.x] = (u16)lane;
  bf16x8 av[8], bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      av[j][i] = (__bf16)(0.001f * (float)(lane + i + j));
      bv[j][i] = (__bf16)(0.002f * (float)((lane ^ i) + j));
    }
    asm volatile("" : "+v"(av[j]), "+v"(bv[j]));
  }
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float fv[8], fc = 1.0001f, spare[4] = {1.f, 2.f, 3.f, 4.f};
  f32x4 vres[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 pv[4], pc = f32x2{1.0001f, 0.9999f};
#pragma unroll
  for (int i = 0; i < 4; ++i) pv[i] = f32x2{0.5f + (float)lane, 0.25f + (float)i};
  unsigned pk[4] = {0, 0, 0, 0};
  uint4 lv[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
  const unsigned lds_a = (unsigned)(uintptr_t)((__attribute__((address_space(3))) u16*)pad) + lane * 16u;
#pragma unroll
  for (int i = 0; i < 8; ++i) fv[i] = 0.5f + (float)lane * 0.001f + (float)i;
#pragma unroll
  for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(spare[i]));
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long t0 = wall_clock64();
  for (int it = 0; it < n_iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        constexpr bool DISTINCT = (MODE & 2) != 0;
        const bf16x8 a = av[DISTINCT ? i : 0], b = bv[DISTINCT ? (i + r) & 7 : 0];
        if (MODE & 1)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
        else
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fv[v & 7]) : "v"(fc));
          else if (KIND == 1) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(fv[v & 7]) : "a"(spare[v & 3]));
          else if (KIND == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(lv[v & 1]) : "v"(lds_a));
          else if (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[v & 3]) : "v"(fv[v & 7]), "v"(fc));
          else if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pv[v & 3]) : "v"(pc));
          else if (KIND == 5) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(pk[v & 3]));
          else if (KIND == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(fv[v & 7]));
          else if (KIND == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(fv[v & 7]) : "v"(fc));
          else if (KIND == 8) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(fv[v & 7]) : "v"(fc));
          else if (KIND == 9) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fv[0]) : "v"(fc));  // ONE dependent chain
          else if (KIND == 10) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fv[v & 1]) : "v"(fc));  // two chains
          else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[0]) : "v"(fv[(v + 1) & 7]), "v"(fc));
        }
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float total = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) total += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + fv[i];
  total += (float)(pk[0] ^ pk[1] ^ pk[2] ^ pk[3]) + (float)(lv[0].x ^ lv[1].y) + vres[0][0] + vres[1][1] + pv[0][0] + pv[1][1] + pv[2][0] + pv[3][1];
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long t1 = wall_clock64();
  if (total == 123.456f) sink[threadIdx.x] = total + (float)pad[(threadIdx.x * 7) & 1023];
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = t1 - t0;
  }
}

template <int WAVES_PER_SIMD, int MODE = 0, int NV = 0, int KIND = 0>
static double clock_probe(const char* label, float* sink) {
  const int blocks = 256 * 8, n_iters = 4096;  // 8 blocks per CU queued behind each other
  const int threads = 256 * WAVES_PER_SIMD > 1024 ? 1024 : 256 * WAVES_PER_SIMD;
  unsigned long long* out;
  CHECK(hipMalloc(&out, (size_t)blocks * 2 * sizeof(unsigned long long)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((clock_probe_kernel<WAVES_PER_SIMD, MODE, NV, KIND>), dim3(blocks), dim3(threads), 0, 0, n_iters, out, sink);
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((clock_probe_kernel<WAVES_PER_SIMD, MODE, NV, KIND>), dim3(blocks), dim3(threads), 0, 0, n_iters, out, sink);
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
  const double mfma_per_wave = (double)n_iters * 32;
  const double waves_per_simd = threads / 256.0;
  const double ghz = cyc / real * 0.1;  // s_memrealtime ticks at 100 MHz
  const double flops = (double)blocks * (threads / 64) * mfma_per_wave * 16 * 16 * 32 * 2;
  printf("%-44s %6.3f GHz shader clock, %5.2f cycles per MFMA per SIMD, %7.1f TFLOP/s by HIP events; ceiling at this clock %7.1f\n",
         label, ghz, cyc / blocks / (mfma_per_wave * waves_per_simd), flops / ms * 1e-9, 1024.0 * 1024.0 * ghz * 1e-3);
  CHECK(hipFree(out));
  return ghz;
}

// STORE: 0 = no global stores; 1 = the epilogue stores 2 x 16 B per lane per chunk and the iteration ends with
// __syncthreads() (fence: s_waitcnt vmcnt(0) covers the just-issued stores); 2 = s_waitcnt vmcnt(0) BEFORE the
// stores (only the weight DMA issued at the top of the iteration is outstanding then) and a bare s_barrier.
template <int VARIANT, int HORNER, bool DMA, bool BARRIER, bool LDS = true, int STORE = 0, int OCC = 2>
__global__ __launch_bounds__(256, OCC) void loop_kernel(const u16* __restrict__ w, int n_chunks, float* __restrict__ sink,
                                                      uint4* __restrict__ out) {
  // OCC = 1: pad the static LDS to 96 KiB so that only one block fits a CU (one wave per SIMD)
  __shared__ __attribute__((aligned(16))) u16 sW[OCC == 1 ? 3 : 2][STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  bf16x8 a_hi[2][KS], a_lo[2][KS];
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u16* src = w + ((size_t)(blockIdx.x & 7) * 16 + mf * 8 + ks) * 1024 + lane * 8;
      a_hi[mf][ks] = *reinterpret_cast<const bf16x8*>(src);
      a_lo[mf][ks] = *reinterpret_cast<const bf16x8*>(src + 512);
    }
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      asm volatile("" : "+v"(a_hi[mf][ks]));
      asm volatile("" : "+v"(a_lo[mf][ks]));
    }

  auto stage_chunk = [&](int chunk, int stage) {
    const u16* src = w + (size_t)chunk * STAGE;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int piece = wave + 4 * u;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 512 + lane * 8),
                                       (__attribute__((address_space(3))) void*)(&sW[stage][piece * 512]), 16, 0, 0);
    }
  };
  stage_chunk(0, 0);
  stage_chunk(1, 1);
  __syncthreads();

  float acc_prev[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc_prev[i] = 0.f;
  float total = 0.f;
  // accumulators persist over the chunks (never re-zeroed), so no chunk's MFMAs are dead code
  f32x4 acc[2][2];
#pragma unroll
  for (int nf = 0; nf < 2; ++nf)
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) acc[nf][mf] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x16 acc2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;

  auto iteration = [&](int c, auto cur_tag) {
    constexpr int cur = decltype(cur_tag)::value;
    if (DMA) stage_chunk(c + 1 < n_chunks ? c + 1 : c, cur ^ 1);
    // dummy epilogue of the previous chunk: HORNER dependent FMAs per accumulator value
    if (HORNER > 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float x = acc_prev[i], pz = 0.3f;
#pragma unroll
        for (int t = 0; t < HORNER; ++t) pz = fmaf(pz, x, 0.25f + 0.125f * t);
        total += pz;
        acc_prev[i] = pz;
      }
    }
    uint4 pend0, pend1;
    if (STORE > 0) {
      pend0 = make_uint4(__float_as_uint(acc_prev[0]), __float_as_uint(acc_prev[1]), __float_as_uint(acc_prev[2]), __float_as_uint(acc_prev[3]));
      pend1 = make_uint4(__float_as_uint(acc_prev[4]), __float_as_uint(acc_prev[5]), __float_as_uint(acc_prev[6]), __float_as_uint(acc_prev[7]));
    }
    uint4* dst = out + ((size_t)(blockIdx.x * 4 + wave) * n_chunks + c) * 128 + lane;
    if (STORE == 1) {
      dst[0] = pend0;
      dst[64] = pend1;
    }
    const u16* st = &sW[cur][lane * 8];
    if (VARIANT == 0) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 wh[2], wl[2];
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
          wh[nf] = LDS ? lds_frag(st + (ks * 2) * 1024 + nf * 512) : a_hi[nf][(ks + 1) % KS];
          wl[nf] = LDS ? lds_frag(st + (ks * 2 + 1) * 1024 + nf * 512) : a_lo[nf][(ks + 1) % KS];
        }
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
          for (int nf = 0; nf < 2; ++nf)
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
              const bf16x8 wv = term == 0 ? wl[nf] : wh[nf];
              const bf16x8 av = term == 1 ? a_lo[mf][ks] : a_hi[mf][ks];
              acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, av, acc[nf][mf], 0, 0, 0);
            }
      }
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc_prev[(nf * 2 + mf) * 4 + r] = acc[nf][mf][r];
    } else {
      // 32x32x16: the same registers reinterpreted as 16 k-steps of 16 (the values are arbitrary): per k-step one
      // hi and one lo weight fragment (1 KiB each), three MFMAs onto two alternating accumulators
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const bf16x8 wh = LDS ? lds_frag(st + kk * 1024) : a_hi[(kk + 1) & 1][((kk + 1) >> 1) % KS];
        const bf16x8 wl = LDS ? lds_frag(st + kk * 1024 + 512) : a_lo[(kk + 1) & 1][((kk + 1) >> 1) % KS];
        const bf16x8 ah = a_hi[kk & 1][kk >> 1], al = a_lo[kk & 1][kk >> 1];
        // the two accumulators strictly alternate: 0 1 0 | 1 0 1 | ...
        const int x = kk & 1;
        acc2[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah, acc2[x], 0, 0, 0);
        acc2[x ^ 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al, acc2[x ^ 1], 0, 0, 0);
        acc2[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah, acc2[x], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_prev[r] = acc2[0][r] + acc2[1][r];
    }
    if (BARRIER && STORE == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the weight DMA issued at the top; last iteration's stores
      dst[0] = pend0;
      dst[64] = pend1;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else if (BARRIER) {
      __syncthreads();
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  for (int c0 = 0; c0 < n_chunks; c0 += 2) {
    iteration(c0, std::integral_constant<int, 0>{});
    iteration(c0 + 1, std::integral_constant<int, 1>{});
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) total += acc_prev[i];
  if (total == 123.456f) sink[tid] = total;
}

template <int VARIANT, int HORNER, bool DMA, bool BARRIER, bool LDS = true, int STORE = 0, int OCC = 2>
void run(const char* label, const u16* w, float* sink, int n_chunks, int blocks) {
  static uint4* out = nullptr;
  if (!out) CHECK(hipMalloc(&out, (size_t)blocks * 4 * n_chunks * 128 * sizeof(uint4)));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((loop_kernel<VARIANT, HORNER, DMA, BARRIER, LDS, STORE, OCC>), dim3(blocks), dim3(256), 0, 0, w, n_chunks, sink, out);
  CHECK(hipEventRecord(a, 0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((loop_kernel<VARIANT, HORNER, DMA, BARRIER, LDS, STORE, OCC>), dim3(blocks), dim3(256), 0, 0, w, n_chunks, sink, out);
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, a, b));
  ms /= reps;
  // MFMA flops: rows (blocks * 128) x 32 features x K 256 x 2 x 3 terms per chunk
  const double flops = (double)blocks * 128 * 32 * 256 * 2 * 3 * n_chunks;
  printf("%-58s %8.3f ms  %7.1f TFLOP/s (MFMA stream)\n", label, ms, flops / ms * 1e-9);
}

// ---------------------------------------------------------------------------------------------------------------
// k-streamed structure (kstream_gemm_kernel / panel_gemm_kernel / phase 1 of the fused kernels): all N = 256 outputs of
// the wave's 32 rows in 32 persistent accumulators (128 VGPRs), per k-step one [256 x 32] weight slab (32 KiB) by DMA and
// this wave's A fragments (hi/lo x 2 row blocks) either fixed registers (AGLOBAL = false) or four 16-byte global loads
// prefetched one k-step ahead (AGLOBAL = true).
template <bool AGLOBAL, bool DMA, bool BARRIER, bool ARESIDENT = false>
__global__ __launch_bounds__(256, 2) void kstream_kernel(const u16* __restrict__ w, const u16* __restrict__ a, int n_ksteps,
                                                         float* __restrict__ sink) {
  __shared__ __attribute__((aligned(16))) u16 sW[2][STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto stage_slab = [&](int ks, int stage) {
    const u16* src = w + (size_t)(ks & 63) * STAGE;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int piece = wave + 4 * u;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 512 + lane * 8),
                                       (__attribute__((address_space(3))) void*)(&sW[stage][piece * 512]), 16, 0, 0);
    }
  };
  // ARESIDENT: every block reads the same few A rows (L2 hits) instead of streaming its own 128 KiB from HBM
  const u16* a_base = a + ((size_t)((ARESIDENT ? (blockIdx.x & 7) : blockIdx.x) * 8 + wave * 2) * n_ksteps * 2) * 512 + lane * 8;
  const size_t a_block = (size_t)n_ksteps * 2 * 512;
  bf16x8 an_hi[2], an_lo[2];
  auto load_a = [&](int ks) {
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      an_hi[mf] = *reinterpret_cast<const bf16x8*>(a_base + mf * a_block + (size_t)ks * 1024);
      an_lo[mf] = *reinterpret_cast<const bf16x8*>(a_base + mf * a_block + (size_t)ks * 1024 + 512);
    }
  };
  f32x4 acc[16][2];
#pragma unroll
  for (int nf = 0; nf < 16; ++nf)
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) acc[nf][mf] = f32x4{0.f, 0.f, 0.f, 0.f};
  stage_slab(0, 0);
  stage_slab(1, 1);
  load_a(0);
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    asm volatile("" : "+v"(an_hi[mf]));
    asm volatile("" : "+v"(an_lo[mf]));
  }
  __syncthreads();
  auto step = [&](int ks, auto cur_tag) {
    constexpr int cur = decltype(cur_tag)::value;
    const int kn = ks + 1 < n_ksteps ? ks + 1 : ks;
    if (DMA) stage_slab(kn, cur ^ 1);
    bf16x8 a_hi[2], a_lo[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      a_hi[mf] = an_hi[mf];
      a_lo[mf] = an_lo[mf];
    }
    if (AGLOBAL) {
      load_a(kn);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int nf = 0; nf < 16; nf += 2) {
      bf16x8 wh[2], wl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        wh[j] = lds_frag(&sW[cur][(nf + j) * 512 + lane * 8]);
        wl[j] = lds_frag(&sW[cur][(16 + nf + j) * 512 + lane * 8]);
      }
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int mf = 0; mf < 2; ++mf)
            acc[nf + j][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(term == 0 ? wl[j] : wh[j], term == 1 ? a_lo[mf] : a_hi[mf],
                                                                      acc[nf + j][mf], 0, 0, 0);
    }
    if (BARRIER) {
      __syncthreads();
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  for (int k0 = 0; k0 < n_ksteps; k0 += 2) {
    step(k0, std::integral_constant<int, 0>{});
    step(k0 + 1, std::integral_constant<int, 1>{});
  }
  float total = 0.f;
#pragma unroll
  for (int nf = 0; nf < 16; ++nf)
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) total += acc[nf][mf][0] + acc[nf][mf][1] + acc[nf][mf][2] + acc[nf][mf][3];
  if (total == 123.456f) sink[tid] = total;
}

template <bool AGLOBAL, bool DMA, bool BARRIER, bool ARESIDENT = false>
void run_kstream(const char* label, const u16* w, const u16* a, float* sink, int n_ksteps, int blocks) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((kstream_kernel<AGLOBAL, DMA, BARRIER, ARESIDENT>), dim3(blocks), dim3(256), 0, 0, w, a, n_ksteps, sink);
  CHECK(hipEventRecord(e0, 0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((kstream_kernel<AGLOBAL, DMA, BARRIER, ARESIDENT>), dim3(blocks), dim3(256), 0, 0, w, a, n_ksteps, sink);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double flops = (double)blocks * 128 * 256 * 32 * 2 * 3 * n_ksteps;
  printf("%-58s %8.3f ms  %7.1f TFLOP/s (MFMA stream)\n", label, ms, flops / ms * 1e-9);
}

// A fragments through a private LDS region instead of VGPR-returning loads: per k-step each wave DMAs its own four
// 1 KiB A pieces (issued BEFORE the weight slab's pieces, so that a counted s_waitcnt can wait for them alone), reads
// them into registers two thirds into the step's MFMAs (the region was consumed a step ago, the data landed a while
// ago), and uses them in the next step.  80 KiB of dynamic LDS per block: two blocks still fit a CU.
template <bool DMA, bool BARRIER>
__global__ __launch_bounds__(256, 2) void kstream_lds_kernel(const u16* __restrict__ w, const u16* __restrict__ a, int n_ksteps,
                                                             float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  u16* sA = smem + 2 * STAGE + (threadIdx.x >> 6) * 2048;  // this wave's 4 pieces [mf][plane][512]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto stage_slab = [&](int ks, int stage) {
    const u16* src = w + (size_t)(ks & 63) * STAGE;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int piece = wave + 4 * u;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 512 + lane * 8),
                                       (__attribute__((address_space(3))) void*)(smem + stage * STAGE + piece * 512), 16, 0, 0);
    }
  };
  const u16* a_base = a + ((size_t)(blockIdx.x * 8 + wave * 2) * n_ksteps * 2) * 512 + lane * 8;
  const size_t a_block = (size_t)n_ksteps * 2 * 512;
  auto stage_a = [&](int ks) {
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(a_base + mf * a_block + (size_t)ks * 1024 + pl * 512),
            (__attribute__((address_space(3))) void*)(sA + (mf * 2 + pl) * 512), 16, 0, 0);
  };
  f32x4 acc[16][2];
#pragma unroll
  for (int nf = 0; nf < 16; ++nf)
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) acc[nf][mf] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a_hi[2], a_lo[2], n_hi[2], n_lo[2];
  stage_a(0);
  stage_slab(0, 0);
  stage_slab(1, 1);
  __syncthreads();
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    n_hi[mf] = lds_frag(sA + (mf * 2) * 512 + lane * 8);
    n_lo[mf] = lds_frag(sA + (mf * 2 + 1) * 512 + lane * 8);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  auto step = [&](int ks, auto cur_tag) {
    constexpr int cur = decltype(cur_tag)::value;
    const int kn = ks + 1 < n_ksteps ? ks + 1 : ks;
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      a_hi[mf] = n_hi[mf];
      a_lo[mf] = n_lo[mf];
    }
    stage_a(kn);                      // 4 pieces, oldest in the queue
    if (DMA) stage_slab(kn, cur ^ 1); // 8 pieces
#pragma unroll
    for (int nf = 0; nf < 16; nf += 2) {
      bf16x8 wh[2], wl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        wh[j] = lds_frag(smem + cur * STAGE + (nf + j) * 512 + lane * 8);
        wl[j] = lds_frag(smem + cur * STAGE + (16 + nf + j) * 512 + lane * 8);
      }
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int mf = 0; mf < 2; ++mf)
            acc[nf + j][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(term == 0 ? wl[j] : wh[j], term == 1 ? a_lo[mf] : a_hi[mf],
                                                                      acc[nf + j][mf], 0, 0, 0);
      if (nf == 10) {
        // two thirds in: this wave's A pieces of the next k-step have landed long ago; only the slab may still fly
        if (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
          n_hi[mf] = lds_frag(sA + (mf * 2) * 512 + lane * 8);
          n_lo[mf] = lds_frag(sA + (mf * 2 + 1) * 512 + lane * 8);
        }
      }
    }
    if (BARRIER) {
      __syncthreads();
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
  };
  for (int k0 = 0; k0 < n_ksteps; k0 += 2) {
    step(k0, std::integral_constant<int, 0>{});
    step(k0 + 1, std::integral_constant<int, 1>{});
  }
  float total = 0.f;
#pragma unroll
  for (int nf = 0; nf < 16; ++nf)
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) total += acc[nf][mf][0] + acc[nf][mf][1] + acc[nf][mf][2] + acc[nf][mf][3];
  if (total == 123.456f) sink[tid] = total;
}

template <bool DMA, bool BARRIER>
void run_kstream_lds(const char* label, const u16* w, const u16* a, float* sink, int n_ksteps, int blocks) {
  const size_t lds = (size_t)(2 * STAGE + 4 * 2048) * sizeof(u16);  // 80 KiB
  CHECK(hipFuncSetAttribute((const void*)kstream_lds_kernel<DMA, BARRIER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((kstream_lds_kernel<DMA, BARRIER>), dim3(blocks), dim3(256), lds, 0, w, a, n_ksteps, sink);
  CHECK(hipEventRecord(e0, 0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((kstream_lds_kernel<DMA, BARRIER>), dim3(blocks), dim3(256), lds, 0, w, a, n_ksteps, sink);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double flops = (double)blocks * 128 * 256 * 32 * 2 * 3 * n_ksteps;
  printf("%-58s %8.3f ms  %7.1f TFLOP/s (MFMA stream)\n", label, ms, flops / ms * 1e-9);
}

int main() {
  const int n_chunks = 64, blocks = 1024;
  std::vector<u16> host((size_t)n_chunks * STAGE);
  for (size_t i = 0; i < host.size(); ++i) host[i] = (u16)(0x3c00 + (i * 2654435761u >> 24));  // ~0.0078..0.03
  u16* w;
  float* sink;
  CHECK(hipMalloc(&w, host.size() * 2));
  CHECK(hipMalloc(&sink, 4096));
  CHECK(hipMemcpy(w, host.data(), host.size() * 2, hipMemcpyHostToDevice));
  printf("clock probe: back-to-back v_mfma_f32_16x16x32_bf16, operands in registers, 2048 blocks\n");
  clock_probe<1>("  1 wave per SIMD ", sink);
  clock_probe<2>("  2 waves per SIMD", sink);
  clock_probe<4>("  4 waves per SIMD", sink);
  clock_probe<1>("  1 wave per SIMD (again, warm)", sink);
  clock_probe<1, 1>("  1 wave per SIMD, AGPR accumulators", sink);
  clock_probe<1, 2>("  1 wave per SIMD, 8 distinct (a, b)", sink);
  clock_probe<1, 3>("  1 wave per SIMD, AGPR acc + distinct", sink);
  clock_probe<2, 3>("  2 waves per SIMD, AGPR acc + distinct", sink);
  printf("one wave per SIMD, AGPR accumulators, N vector instructions after every MFMA:\n");
  clock_probe<1, 1, 1, 0>("  1 v_fma_f32 / MFMA", sink);
  clock_probe<1, 1, 2, 0>("  2 v_fma_f32 / MFMA", sink);
  clock_probe<1, 1, 3, 0>("  3 v_fma_f32 / MFMA", sink);
  clock_probe<1, 1, 4, 0>("  4 v_fma_f32 / MFMA", sink);
  clock_probe<1, 1, 6, 0>("  6 v_fma_f32 / MFMA", sink);
  clock_probe<1, 1, 8, 0>("  8 v_fma_f32 / MFMA", sink);
  clock_probe<1, 1, 2, 1>("  2 v_accvgpr_read / MFMA", sink);
  clock_probe<1, 1, 4, 1>("  4 v_accvgpr_read / MFMA", sink);
  clock_probe<1, 1, 2, 3>("  2 v_cvt_pk_bf16_f32 / MFMA", sink);
  clock_probe<1, 1, 4, 3>("  4 v_cvt_pk_bf16_f32 / MFMA", sink);
  clock_probe<1, 1, 1, 2>("  1 ds_read_b128 / MFMA", sink);
  clock_probe<1, 1, 2, 2>("  2 ds_read_b128 / MFMA", sink);
  // (an MFMA with C in AGPRs and D in VGPRs does not assemble: one acc_cd bit selects the class of both)
  clock_probe<1, 1, 2, 4>("  2 v_pk_fma_f32 / MFMA", sink);
  clock_probe<1, 1, 4, 4>("  4 v_pk_fma_f32 / MFMA", sink);
  clock_probe<1, 1, 4, 5>("  4 v_and_b32 (literal) / MFMA", sink);
  clock_probe<1, 1, 2, 6>("  2 v_exp_f32 / MFMA", sink);
  clock_probe<1, 1, 4, 6>("  4 v_exp_f32 / MFMA", sink);
  clock_probe<1, 1, 4, 7>("  4 v_mul_f32 / MFMA", sink);
  clock_probe<1, 1, 4, 8>("  4 v_sub_f32 / MFMA", sink);
  clock_probe<1, 1, 2, 9>("  2 DEPENDENT v_fma_f32 / MFMA (one chain)", sink);
  clock_probe<1, 1, 3, 9>("  3 DEPENDENT v_fma_f32 / MFMA (one chain)", sink);
  clock_probe<1, 1, 4, 9>("  4 DEPENDENT v_fma_f32 / MFMA (one chain)", sink);
  clock_probe<1, 1, 4, 10>("  4 v_fma_f32 / MFMA in two chains", sink);
  clock_probe<2, 1, 4, 9>("  2 waves/SIMD: 4 DEPENDENT v_fma_f32 / MFMA", sink);
  clock_probe<2, 1, 4, 0>("  2 waves/SIMD: 4 v_fma_f32 / MFMA", sink);
  clock_probe<2, 1, 8, 0>("  2 waves/SIMD: 8 v_fma_f32 / MFMA", sink);
  printf("1024 blocks x 4 waves x 32 rows, K=256 bf16x3, 64 chunks of 32 features (one Wi GEMM of ModernBERT-xsmall)\n");
#define CASES(V, NAME)                                                                        \
  run<V, 0, false, false, false>(NAME "  registers only (no LDS reads)     ", w, sink, n_chunks, blocks); \
  run<V, 0, false, false, false>(NAME "  registers only (no LDS reads) again", w, sink, n_chunks, blocks); \
  run<V, 8, false, false, false>(NAME "  registers only, epilogue 16x8 fma ", w, sink, n_chunks, blocks); \
  run<V, 0, false, false>(NAME "  no dma, no barrier  epilogue 0", w, sink, n_chunks, blocks);     \
  run<V, 0, true, false>(NAME "  dma,    no barrier  epilogue 0", w, sink, n_chunks, blocks);      \
  run<V, 0, true, true>(NAME "  dma,    barrier     epilogue 0", w, sink, n_chunks, blocks);       \
  run<V, 2, true, true>(NAME "  dma,    barrier     epilogue 16x2 fma", w, sink, n_chunks, blocks); \
  run<V, 4, true, true>(NAME "  dma,    barrier     epilogue 16x4 fma", w, sink, n_chunks, blocks); \
  run<V, 8, true, true>(NAME "  dma,    barrier     epilogue 16x8 fma", w, sink, n_chunks, blocks); \
  run<V, 8, false, false>(NAME "  no dma, no barrier  epilogue 16x8 fma", w, sink, n_chunks, blocks); \
  run<V, 8, true, true, true, 1>(NAME "  dma, __syncthreads  epilogue 16x8 fma + stores", w, sink, n_chunks, blocks); \
  run<V, 8, true, true, true, 2>(NAME "  dma, wait-then-store + bare barrier, 16x8 fma", w, sink, n_chunks, blocks); \
  run<V, 0, false, false, false, 0, 1>(NAME "  1 wave/SIMD: registers only", w, sink, n_chunks, blocks); \
  run<V, 0, false, false, true, 0, 1>(NAME "  1 wave/SIMD: LDS reads, no dma, no barrier", w, sink, n_chunks, blocks); \
  run<V, 0, true, true, true, 0, 1>(NAME "  1 wave/SIMD: dma, barrier, epilogue 0", w, sink, n_chunks, blocks); \
  run<V, 8, true, true, true, 0, 1>(NAME "  1 wave/SIMD: dma, barrier, epilogue 16x8 fma", w, sink, n_chunks, blocks);
  CASES(0, "16x16x32")
  CASES(1, "32x32x16")
  {
    // k-streamed structure: K = 1024 (32 k-steps), the MLP output projection of ModernBERT-xsmall
    const int n_ksteps = 32;
    u16* a;
    CHECK(hipMalloc(&a, (size_t)blocks * 8 * n_ksteps * 2 * 512 * 2));
    CHECK(hipMemset(a, 0x3c, (size_t)blocks * 8 * n_ksteps * 2 * 512 * 2));
    printf("k-streamed structure, 32 accumulator fragments per wave, K = 1024:\n");
    run_kstream<false, false, false>("kstream  A in registers, no dma, no barrier", w, a, sink, n_ksteps, blocks);
    run_kstream<false, true, true>("kstream  A in registers, dma, barrier", w, a, sink, n_ksteps, blocks);
    run_kstream<true, true, true>("kstream  A from global (prefetch 1), dma, barrier", w, a, sink, n_ksteps, blocks);
    run_kstream<true, false, false>("kstream  A from global (prefetch 1), no dma, no barrier", w, a, sink, n_ksteps, blocks);
    run_kstream<true, true, true, true>("kstream  A from L2 (same rows for all blocks), dma, barrier", w, a, sink, n_ksteps, blocks);
    run_kstream_lds<true, true>("kstream  A through private LDS (DMA), dma, barrier", w, a, sink, n_ksteps, blocks);
    run_kstream<true, true, true>("kstream  A from global (prefetch 1), dma, barrier  [again]", w, a, sink, n_ksteps, blocks);
    run_kstream<false, true, true>("kstream  A in registers, dma, barrier  [again]", w, a, sink, n_ksteps, blocks);
  }
  return 0;
}
