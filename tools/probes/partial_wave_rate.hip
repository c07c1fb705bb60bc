// Does a wavefront with part of its lanes switched off issue vector instructions faster?  (If the SIMD skipped the
// 16-lane passes whose lanes are all inactive, a half-empty wavefront would cost half the issue cycles, and small
// batches could be spread over twice the CUs.)  Same instruction stream, 1 024 wavefronts per launch x 4 per SIMD;
// active lanes 64, 32 (low half), 16 (first quarter), 32 (every other lane).
// hipcc --offload-arch=gfx950 -O3 -o partial_wave_rate partial_wave_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(256) rate_kernel(uint32_t* out, int mode, int iters) {
  const uint32_t lane = threadIdx.x & 63u;
  const bool on = mode == 0 ? true : mode == 1 ? lane < 32 : mode == 2 ? lane < 16 : (lane & 1u) == 0;
  uint32_t a0 = threadIdx.x * 2654435761u + 1, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77, a3 = a1 + 1234567;
  const uint32_t m = 0x87654321u | blockIdx.x;
  if (on) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int k = 0; k < 16; k++) {
        a0 = __umulhi(a0, m) + k;
        a1 = __umulhi(a1, m) + k;
        a2 = __umulhi(a2, m) + k;
        a3 = __umulhi(a3, m) + k;
      }
    }
  }
  if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345u) out[threadIdx.x] = a0;  // keep the chains alive
}

int main() {
  uint32_t* out;
  hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4000;
  const char* names[4] = {"64 lanes", "lanes 0..31", "lanes 0..15", "even lanes"};
  for (int rep = 0; rep < 2; rep++)
    for (int mode = 0; mode < 4; mode++) {
      hipLaunchKernelGGL(rate_kernel, dim3(1024), dim3(256), 0, 0, out, mode, 10);
      hipEventRecord(e0);
      hipLaunchKernelGGL(rate_kernel, dim3(1024), dim3(256), 0, 0, out, mode, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%-12s %8.3f ms\n", names[mode], ms);
    }
  return 0;
}
