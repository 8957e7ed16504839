// Microbenchmark (measurement aid): does the immediate offset of global_load_lds_dwordx4 move the GLOBAL address, the LDS
// address, or both?  One wave copies piece 0 with offset 0 and "piece 1" with offset 1024 through the same M0 / pointer.
//   hipcc --offload-arch=gfx950 -O3 -o dma_offset_probe dma_offset_probe.hip && ./dma_offset_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* src, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[1024];  // 4 KiB
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 4),
                                   (__attribute__((address_space(3))) void*)(&lds[0]), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 4),
                                   (__attribute__((address_space(3))) void*)(&lds[0]), 16, 1024, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<unsigned> h(2048);
  for (int i = 0; i < 2048; ++i) h[i] = i;
  unsigned *src, *out;
  hipMalloc(&src, 8192); hipMalloc(&out, 4096);
  hipMemcpy(src, h.data(), 8192, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out);
  std::vector<unsigned> o(1024);
  hipMemcpy(o.data(), out, 4096, hipMemcpyDeviceToHost);
  printf("lds dword 0..3: %u %u %u %u | dword 256..259 (byte 1024): %u %u %u %u | dword 512: %x\n", o[0], o[1], o[2], o[3], o[256], o[257], o[258], o[259], o[512]);
  printf("expected if the offset moves BOTH addresses: 0 1 2 3 | 256 257 258 259\n");
  return 0;
}
