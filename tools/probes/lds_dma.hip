// Probe: buffer_load_dwordx4 ... lds (global -> LDS without registers) on gfx950.
// Each wave copies its 1 KiB row of `src` into LDS row (wave + shift) and reads it back with ds_read_b128.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
extern __shared__ uint4 lds[];
__global__ void __launch_bounds__(256) k(const uint32_t* src, uint32_t src_bytes, uint32_t* out, uint32_t row_shift) {
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t lds_base = (uint32_t)(uintptr_t)lds;
  const uint32_t row = lds_base + ((wave + row_shift) & 3) * 1024;  // wave-uniform LDS byte address
  const uint32_t voff = blockIdx.x * 4096 + wave * 1024 + lane * 16;
  const uint64_t base = (uint64_t)(uintptr_t)src;
  uint4 got;
  const uint32_t rd = row + lane * 16;
  asm volatile(
      "s_mov_b64 s[48:49], %[base]\n"
      "s_and_b32 s49, s49, 0xffff\n"
      "s_mov_b32 s50, %[bytes]\n"
      "s_mov_b32 s51, 0x00020000\n"
      "s_mov_b32 m0, %[row]\n"
      "s_nop 0\n"
      "buffer_load_dwordx4 %[voff], s[48:51], 0 offen lds\n"
      "s_waitcnt vmcnt(0)\n"
      "ds_read_b128 %[got], %[rd]\n"
      "s_waitcnt lgkmcnt(0)\n"
      : [got] "=&v"(got)
      : [base] "s"(base), [bytes] "s"(src_bytes), [row] "s"(row), [voff] "v"(voff), [rd] "v"(rd)
      : "s48", "s49", "s50", "s51", "memory");
  uint32_t* o = out + (blockIdx.x * 256 + threadIdx.x) * 4;
  o[0] = got.x; o[1] = got.y; o[2] = got.z; o[3] = got.w;
}
int main() {
  const int blocks = 64, n = blocks * 256 * 4;
  uint32_t *src, *out; CK(hipMalloc(&src, n * 4)); CK(hipMalloc(&out, n * 4));
  uint32_t* h = (uint32_t*)malloc(n * 4); uint32_t* g = (uint32_t*)malloc(n * 4);
  for (int i = 0; i < n; i++) h[i] = 0x9e3779b9u * (uint32_t)(i + 1);
  CK(hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice));
  for (uint32_t shift = 0; shift < 2; shift++) {
    CK(hipMemset(out, 0, n * 4));
    k<<<blocks, 256, 4096>>>(src, (uint32_t)(n * 4), out, shift);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(g, out, n * 4, hipMemcpyDeviceToHost));
    long bad = 0; for (int i = 0; i < n; i++) bad += g[i] != h[i];
    printf("row shift %u: %ld of %d words differ (first got %08x want %08x)\n", shift, bad, n, g[0], h[0]);
  }
  return 0;
}
