// cu_mask_probe.hip -- which physical CUs a hipExtStreamCreateWithCUMask stream runs on (gfx950, 8 XCDs x 32 CUs).
// Every block records HW_ID / XCC_ID; the host prints, per mask pattern, how many distinct (XCC, SE, CU) the blocks ran on and the
// per-XCC counts.  Patterns: contiguous halves of the bit index, alternate bits, runs of 4 / 8 / 32 bits.
// build: hipcc --offload-arch=gfx950 -O2 microbench/cu_mask_probe.hip -o microbench/cu_mask_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <map>
#include <set>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                       \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

__global__ void where_kernel(unsigned* out, unsigned long long spin_ticks) {
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[blockIdx.x * 2] = hw;
    out[blockIdx.x * 2 + 1] = xcc;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(4);  // keep the CU busy so that the grid spreads
  }
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cus = prop.multiProcessorCount, words = (n_cus + 31) / 32;
  const int blocks = 2048;
  unsigned* d_out;
  CHECK(hipMalloc(&d_out, blocks * 2 * sizeof(unsigned)));
  std::vector<unsigned> h(blocks * 2);
  struct Pattern { const char* name; int group; };  // group g > 0: bit c belongs to part (c / g) % 2; 0: contiguous halves
  const Pattern patterns[] = {{"contiguous halves", 0}, {"alternate bits", 1}, {"runs of 4", 4}, {"runs of 8", 8}, {"runs of 32", 32}};
  for (const Pattern& pt : patterns)
    for (int part = 0; part < 2; ++part) {
      std::vector<uint32_t> mask(words, 0u);
      for (int c = 0; c < n_cus; ++c)
        if ((pt.group > 0 ? (c / pt.group) % 2 : c * 2 / n_cus) == part) mask[c / 32] |= 1u << (c % 32);
      hipStream_t st;
      CHECK(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask.data()));
      CHECK(hipMemsetAsync(d_out, 0xff, blocks * 2 * sizeof(unsigned), st));
      hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(512), 65536, st, d_out, 2000ull);  // 64 KiB LDS: one block per CU at a time... two
      CHECK(hipStreamSynchronize(st));
      CHECK(hipMemcpy(h.data(), d_out, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
      std::set<unsigned> cus;
      std::map<unsigned, std::set<unsigned>> per_xcc;
      for (int b = 0; b < blocks; ++b) {
        const unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;  // gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
        const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        cus.insert(key);
        per_xcc[xcc].insert(key & 0xfff);
      }
      printf("%-18s part %d: %3zu distinct CUs;  per XCC:", pt.name, part, cus.size());
      for (auto& kv : per_xcc) printf(" x%u=%zu", kv.first, kv.second.size());
      printf("\n");
      if (pt.group == 1 || pt.group == 0) {
        printf("    XCC 0 CUs (se.sh.cu):");
        for (unsigned k : per_xcc[0]) printf(" %u.%u.%u", (k >> 8) & 7, (k >> 4) & 1, k & 0xf);
        printf("\n");
      }
      CHECK(hipStreamDestroy(st));
    }
  return 0;
}
