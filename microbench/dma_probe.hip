// Microbenchmark (measurement aid, not product code): what bounds the operand streams of the GEMM kernels -- the
// per-CU load THROUGHPUT or the bytes IN FLIGHT (latency x occupancy)?  Every block loops: request a stage of S KiB
// (global -> LDS DMA, 16 B per lane, or register loads), keep D stages in flight, wait for the oldest, barrier.
//   source "shared":  every block streams the same F MiB window (a layer's weight panels: L2 / MALL hits)
//   source "private": every block streams its own rows (activations: HBM)
// Reported: bytes per shader clock per CU and aggregate TB/s, over S, D, blocks per CU and the window.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o dma_probe dma_probe.hip && ./dma_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                     \
  do {                                                               \
    hipError_t e_ = (x);                                             \
    if (e_ != hipSuccess) {                                          \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); \
      exit(1);                                                       \
    }                                                                \
  } while (0)

typedef unsigned short u16;

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// S_KIB per stage, DEPTH stages in flight, DMA: global_load_lds, else global_load_dwordx4 into registers
template <int S_KIB, int DEPTH, bool DMA>
__global__ __launch_bounds__(256) void stream_kernel(const u16* __restrict__ src, size_t window_elems, size_t block_stride_elems,
                                                     int iters, unsigned long long* __restrict__ cycles, uint4* __restrict__ sink) {
  constexpr int PER_WAVE = S_KIB / 4;  // 1 KiB requests per wave and stage
  static_assert(PER_WAVE * DEPTH <= 48, "vmcnt range");
  __shared__ __attribute__((aligned(16))) u16 lds[(DEPTH + 1) * S_KIB * 512];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t base = (size_t)blockIdx.x * block_stride_elems;
  uint4 acc = make_uint4(0, 0, 0, 0);
  auto request = [&](int it) {
    const size_t off = (base + (size_t)it * (S_KIB * 512)) % window_elems;
#pragma unroll
    for (int u = 0; u < PER_WAVE; ++u) {
      const int piece = wave + 4 * u;
      const u16* s = src + off + (size_t)piece * 512 + lane * 8;
      if (DMA) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                         (__attribute__((address_space(3))) void*)(&lds[((it % (DEPTH + 1)) * S_KIB + piece) * 512]), 16, 0, 0);
      } else {
        const uint4 v = *reinterpret_cast<const uint4*>(s);
        // consumed after the wait below: the registers of a stage stay live for DEPTH iterations in a real kernel; here
        // the xor is issued right away and the waitcnt the compiler inserts is what we pay -- so register loads are only
        // measured with DEPTH = 1 semantics (see main)
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
      }
    }
  };
  const unsigned long long c0 = __builtin_readcyclecounter();
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) request(d);
  for (int it = 0; it < iters; ++it) {
    request(it + DEPTH);
    if (DMA) wait_vm<PER_WAVE * DEPTH>();
    __builtin_amdgcn_s_barrier();  // (no fence: the compiler would wait for EVERY outstanding DMA before an LDS read)
    if (DMA) {  // touch the landed stage (one 16-byte read per lane), as a consumer would
      uint4 v;
      const uint32_t a = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) u16*)&lds[0]) +
                         (uint32_t)((((it % (DEPTH + 1)) * S_KIB + wave) * 512 + lane * 8) * 2);
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
      __builtin_amdgcn_s_barrier();
    }
  }
  wait_vm<0>();
  const unsigned long long c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = c1 - c0;
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345679u) sink[blockIdx.x] = acc;
}

struct Result {
  double bytes_per_clk_cu, tbps, ms;
};

template <int S_KIB, int DEPTH, bool DMA>
Result run(const u16* src, size_t window_elems, bool shared_window, int blocks_per_cu, int iters, unsigned long long* cyc_dev,
           uint4* sink) {
  const int n_cu = 256;
  const int blocks = n_cu * blocks_per_cu;
  const size_t stride = shared_window ? 0 : (size_t)(iters + DEPTH) * S_KIB * 512;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {  // second run timed
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((stream_kernel<S_KIB, DEPTH, DMA>), dim3(blocks), dim3(256), 0, 0, src, window_elems, stride, iters, cyc_dev, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> cyc(blocks);
  CHECK(hipMemcpy(cyc.data(), cyc_dev, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double mean = 0;
  for (auto c : cyc) mean += (double)c;
  mean /= blocks;
  const double bytes_block = (double)(iters + DEPTH) * S_KIB * 1024;
  Result r;
  r.bytes_per_clk_cu = bytes_block * blocks_per_cu / mean;
  r.tbps = bytes_block * blocks / (ms * 1e-3) / 1e12;
  r.ms = ms;
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return r;
}

int main() {
  const size_t big = (size_t)6 << 30;  // bytes
  u16* src;
  CHECK(hipMalloc(&src, big));
  CHECK(hipMemset(src, 1, big));
  unsigned long long* cyc;
  uint4* sink;
  CHECK(hipMalloc(&cyc, 4096 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&sink, 4096 * sizeof(uint4)));
  const int iters = 2000;
  printf("%-10s %-8s %6s %5s %7s | %10s %8s %8s\n", "path", "source", "S KiB", "depth", "blk/CU", "B/clk/CU", "TB/s", "ms");
  auto line = [&](const char* path, const char* source, int s, int d, int b, Result r) {
    printf("%-10s %-8s %6d %5d %7d | %10.2f %8.2f %8.3f\n", path, source, s, d, b, r.bytes_per_clk_cu, r.tbps, r.ms);
    fflush(stdout);
  };
  struct Src { const char* name; size_t window; bool shared; };
  const Src sources[] = {{"L2 2MiB", (size_t)2 << 20, true}, {"MALL 64M", (size_t)64 << 20, true}, {"HBM priv", big, false}};
  for (const Src& s : sources) {
    const size_t w = s.window / 2;
    for (int b : {1, 2}) {
      line("dma", s.name, 16, 1, b, run<16, 1, true>(src, w, s.shared, b, iters, cyc, sink));
      line("dma", s.name, 16, 2, b, run<16, 2, true>(src, w, s.shared, b, iters, cyc, sink));
      line("dma", s.name, 16, 3, b, run<16, 3, true>(src, w, s.shared, b, iters, cyc, sink));
      line("dma", s.name, 24, 1, b, run<24, 1, true>(src, w, s.shared, b, iters, cyc, sink));
      line("dma", s.name, 24, 2, b, run<24, 2, true>(src, w, s.shared, b, iters, cyc, sink));
      line("dma", s.name, 32, 1, b, run<32, 1, true>(src, w, s.shared, b, iters, cyc, sink));
      if (b == 1) {
        line("dma", s.name, 32, 2, b, run<32, 2, true>(src, w, s.shared, b, iters, cyc, sink));
        line("dma", s.name, 32, 3, b, run<32, 3, true>(src, w, s.shared, b, iters, cyc, sink));
        line("dma", s.name, 48, 1, b, run<48, 1, true>(src, w, s.shared, b, iters, cyc, sink));
        line("dma", s.name, 48, 2, b, run<48, 2, true>(src, w, s.shared, b, iters, cyc, sink));
        line("dma", s.name, 64, 1, b, run<64, 1, true>(src, w, s.shared, b, iters, cyc, sink));
      }
      line("regs", s.name, 16, 1, b, run<16, 1, false>(src, w, s.shared, b, iters, cyc, sink));
      line("regs", s.name, 32, 1, b, run<32, 1, false>(src, w, s.shared, b, iters, cyc, sink));
    }
  }
  return 0;
}
