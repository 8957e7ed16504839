// Measurement aid (round 4): can a 128-row block of the whole-layer kernel carry a FIFTH, thin "loader" wave?
// A kernel's register allocation is one number in its descriptor -- every wave of every block gets the same VGPR + AGPR
// allocation -- and a SIMD's file has 512 registers per lane.  A 320-thread block puts two waves on one of the CU's four
// SIMDs, so hipcc caps such a kernel at 256 registers per lane: the same body that needs ~400 live registers compiles
// spill-free under __launch_bounds__(256) (one wave per SIMD) and spills to scratch under __launch_bounds__(320).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Rpass-analysis=kernel-resource-usage -c loader_wave_probe.hip -o /dev/null
// (CPU-only: the remarks are the result; recorded in profiles/r04_loader_wave_probe.txt)
#include <hip/hip_runtime.h>

template <int THREADS>
__global__ __launch_bounds__(THREADS) void hungry_kernel(const float* __restrict__ in, float* __restrict__ out) {
  constexpr int N = 400;  // live values per lane, as the whole-layer kernel's accumulators + operands
  float v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = in[threadIdx.x + 320 * i];
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));  // all of them live at once
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i) s += v[i] * (float)(i + 1);
  out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template __global__ void hungry_kernel<256>(const float*, float*);
template __global__ void hungry_kernel<320>(const float*, float*);
