// Fr multiplication rate as a function of waves per SIMD (register-only chain)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../zkir_amd/csrc/hip/fr32.cuh"
using namespace zkir;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed, int iters) {
  Fr x = fr_one_mont(), y = fr_r2();
  x.l[0] ^= threadIdx.x + seed; x.l[7] &= 0x0fffffffu;
  y.l[0] ^= blockIdx.x; y.l[7] &= 0x0fffffffu;
  for (int it = 0; it < iters; it++) { x = fr_mul(x, y); y = fr_mul(y, x); }
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s ^= x.l[i] ^ y.l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  uint32_t* buf; CK(hipMalloc(&buf, 256 * 8 * 8 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000;
  for (int wps : {1, 2, 3, 4, 8}) {
    const int blocks = 256 * wps;  // 4 waves per block: one per SIMD of a CU
    k<<<blocks, 256>>>(buf, 1, 10); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); k<<<blocks, 256>>>(buf, 1, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double rate = (double)blocks * 256 * 2.0 * iters / (ms * 1e-3);
    printf("waves/SIMD %d: %.1f G Fr-mul/s (%.2f ms)\n", wps, rate / 1e9, ms);
  }
  return 0;
}
