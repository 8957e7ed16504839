// opk_kstream.hip.h -- fragment-packed activations and the k-streamed output projection (kstream_gemm_kernel); last part of
// what opk_rowgemm.hip.h provides
#pragma once

#include <type_traits>
#include <utility>

#include "opk_common.hip.h"
#include "opk_rowgemm_pack.hip.h"

namespace opk {

// ----------------------------------------------------------------------------------------------
// Fragment-packed activations.  An activation matrix [rows x C] that is consumed as the MFMA operand
// of the next GEMM is stored as 1 KiB pieces  [row/16][C/32][plane][lane = 16*(k%32/8) + row%16][8 k]:
// exactly one wave-instruction of 16-byte lanes, in lane order.  Producer epilogues store whole pieces
// (one fully coalesced 1 KiB store per wave), consumers load their fragment with one fully coalesced
// 1 KiB load straight into registers -- no LDS staging, no row-strided 8-byte accesses.
// ----------------------------------------------------------------------------------------------

#ifdef OPK_PACK_KERNELS  // weight re-packing runs in op_api.hip only
// dst[ks][plane][nf][g][i][e] <- W[nf*16 + i][ks*32 + g*8 + e]   (W is [N][K]; chunk = one k-step of all N)
__global__ void pack_kstream_kernel(const float* __restrict__ src, int N, int K, int permute, u16* __restrict__ dst,
                                    int zero_lo, int* __restrict__ any_lo, int f16 = 0) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)N * K) return;
  const int NF = N / 16;
  size_t t = idx;
  const int e = (int)(t & 7); t >>= 3;
  const int i = (int)(t & 15); t >>= 4;
  const int g = (int)(t & 3); t >>= 2;
  const int nf = (int)(t % NF);
  const int ks = (int)(t / NF);
  // permute: accumulator slot (nf, i = 4g' + r) holds output feature 32(nf>>1) + 8g' + 4(nf&1) + r, so that the
  // accumulators of fragments (2s, 2s+1) ARE the 8 k-values of lane slot g' of k-step s of the next GEMM.
  const int row = permute ? (32 * (nf >> 1) + 8 * (i >> 2) + 4 * (nf & 1) + (i & 3)) : (nf * 16 + i);
  const float v = src[(size_t)row * K + ks * 32 + g * 8 + e];
  const size_t base = ((size_t)ks * 2 * NF + nf) * 512 + (size_t)g * 128 + i * 8 + e;
  if (f16) {  // kernel set "f16": fp16 hi plane, zero lo plane
    dst[base] = f2h(v);
    dst[base + (size_t)NF * 512] = (u16)0;
    return;
  }
  const u16 h = f2bf(v);
  const u16 l = f2bf(v - bf2f(h));
  if ((l & 0x7fffu) != 0) *any_lo = 1;
  dst[base] = h;
  dst[base + (size_t)NF * 512] = zero_lo ? (u16)0 : l;
}
#endif

struct KStreamParams {
  const u16* a_fp;  // fragment-packed activations [r_pad/16][n_ksteps][2 planes][512]
  const u16* wp;    // packed weights [n_ksteps][2 planes][NF][512]
  int n_ksteps;     // K / 32
  float* x;         // fp32 [r_pad][N], x += A W^T
};

// x[128 or 256 rows, N = 16*NF] += A[rows, K] W[N, K]^T with K streamed: per k-step the block DMAs one
// [N x 32] weight slab into LDS (double-buffered) while every wave pulls its own two A fragments straight
// from the fragment-packed activation (prefetched one k-step ahead) and keeps all N outputs of its 32 rows
// in accumulators (NF x 2 x 4 registers).
template <int NF, int T, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void kstream_gemm_kernel(KStreamParams p) {
  constexpr bool W_LO = (T & T_RIGHT_LO) != 0, A_LO = (T & T_LEFT_LO) != 0;
  constexpr int PLANES = W_LO ? 2 : 1;
  constexpr int STAGE = NF * PLANES * 512;        // elements per LDS stage
  constexpr int CHUNK_SRC = NF * 2 * 512;         // elements per k-step in the packed weights
  constexpr int WAVE_PIECES = STAGE / (WAVES * 512);
  static_assert(STAGE % (WAVES * 512) == 0, "stage must split evenly over the waves");
  __shared__ __attribute__((aligned(16))) u16 sW[2][STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15;
  const int g = lane >> 4;
  const int m0 = blockIdx.x * (WAVES * 32) + wave * 32;
  const int nks = p.n_ksteps;

  auto stage_chunk = [&](int ks, int stage) {
    const u16* src = p.wp + (size_t)ks * CHUNK_SRC;
#pragma unroll
    for (int u = 0; u < WAVE_PIECES; ++u) {
      const int piece = wave + WAVES * u;  // stage = [plane][nf] pieces; source = same order (2 planes)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 512 + lane * 8),
                                       (__attribute__((address_space(3))) void*)(&sW[stage][piece * 512]), 16, 0, 0);
    }
  };
  // A fragments of k-step ks: piece (rb, ks, plane) of the fragment-packed activation, 16 bytes per lane
  const u16* a_base0 = p.a_fp + ((size_t)(m0 >> 4) * nks * 2) * 512 + lane * 8;
  const u16* a_base1 = a_base0 + (size_t)nks * 2 * 512;
  bf16x8 an_hi[2], an_lo[2];
  auto load_a = [&](int ks) {
    an_hi[0] = *reinterpret_cast<const bf16x8*>(a_base0 + (size_t)ks * 1024);
    an_hi[1] = *reinterpret_cast<const bf16x8*>(a_base1 + (size_t)ks * 1024);
    if (A_LO) {
      an_lo[0] = *reinterpret_cast<const bf16x8*>(a_base0 + (size_t)ks * 1024 + 512);
      an_lo[1] = *reinterpret_cast<const bf16x8*>(a_base1 + (size_t)ks * 1024 + 512);
    }
  };

  f32x4 acc[NF][2];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) acc[nf][mf] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage_chunk(0, 0);
  load_a(0);
  // Retire the first fragment loads HERE (empty asm "rewrites" the registers): a load still pending at the loop
  // header makes the compiler drain everything (vmcnt(0)) right after the loop body has issued its DMA.
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    asm volatile("" : "+v"(an_hi[mf]));
    if (A_LO) asm volatile("" : "+v"(an_lo[mf]));
  }
  __syncthreads();

  for (int k0 = 0; k0 < nks; k0 += 2) {
#pragma unroll
    for (int cur = 0; cur < 2; ++cur) {
      const int ks = k0 + cur;
      if (ks >= nks) break;
      const int kn = ks + 1 < nks ? ks + 1 : ks;
      stage_chunk(kn, cur ^ 1);
      bf16x8 a_hi[2], a_lo[2];
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) {
        a_hi[mf] = an_hi[mf];
        a_lo[mf] = an_lo[mf];
      }
      load_a(kn);                              // prefetch the next k-step's fragments ...
      __builtin_amdgcn_sched_barrier(0);       // ... and keep the loads up here, ahead of the MFMAs
#pragma unroll
      for (int nf = 0; nf < NF; nf += 2) {  // term-major over 2 fragments x 2 row blocks (see rowgemm_kernel phase 1)
        bf16x8 wh[2], wl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          wh[j] = lds_frag(&sW[cur][(nf + j) * 512 + lane * 8]);
          wl[j] = W_LO ? lds_frag(&sW[cur][(NF + nf + j) * 512 + lane * 8]) : wh[j];
        }
#pragma unroll
        for (int term = 0; term < 3; ++term) {
          if ((term == 0 && !W_LO) || (term == 1 && !A_LO)) continue;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
              acc[nf + j][mf] = mfma16(term == 0 ? wl[j] : wh[j], term == 1 ? a_lo[mf] : a_hi[mf], acc[nf + j][mf]);
        }
      }
      __syncthreads();
    }
  }

  // x += acc : the weights are packed with permuted output features (pack_kstream_kernel), accumulator slot
  // (nf, g, r) is feature 32(nf>>1) + 8g + 4(nf&1) + r of token m0 + 16mf + l15
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    float* xrow = p.x + (size_t)(m0 + mf * 16 + l15) * (NF * 16) + g * 8;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      float4* px = reinterpret_cast<float4*>(xrow + 32 * (nf >> 1) + 4 * (nf & 1));
      float4 r4 = load_stream_f4(reinterpret_cast<const float*>(px));
      r4.x += acc[nf][mf][0];
      r4.y += acc[nf][mf][1];
      r4.z += acc[nf][mf][2];
      r4.w += acc[nf][mf][3];
      store_stream16(reinterpret_cast<float*>(px), r4);
    }
  }
}

}  // namespace opk
