// How many 256-thread blocks with N KiB of dynamic LDS fit one gfx950 CU?  (160 KiB LDS per CU.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 2) void k(float* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  lds[threadIdx.x] = (unsigned char)threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = lds[5];
}
int main() {
  for (int kib : {32, 64, 72, 78, 79, 80, 81, 96, 128, 160}) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kib * 1024);
    int blocks = -1;
    hipError_t e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k, 256, (size_t)kib * 1024);
    float* out; hipMalloc(&out, 4096);
    hipLaunchKernelGGL(k, dim3(8), dim3(256), (size_t)kib * 1024, 0, out);
    hipError_t e3 = hipDeviceSynchronize();
    printf("%3d KiB: setattr=%s occupancy=%d (%s) launch=%s\n", kib, hipGetErrorName(e), blocks, hipGetErrorName(e2), hipGetErrorName(e3));
    hipFree(out);
  }
  return 0;
}
