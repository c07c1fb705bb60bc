// dependent-issue latencies of the integer VALU instructions at one wave per SIMD
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed, int iters) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 ^ 5, a7 = a0 ^ 9;
  uint32_t b = seed | 1u, c = seed * 3u + 7u;
  uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3;
  uint64_t sink;
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {  // independent adds, 8 chains
      asm volatile(R4("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                      "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if (MODE == 1) {  // dependent adds
      asm volatile(R16("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n") : "+v"(a0) : "v"(b));
    } else if (MODE == 2) {  // dependent 64-bit multiply-adds (through the addend)
      asm volatile(R16("v_mad_u64_u32 %0, %1, %2, %3, %0\n v_mad_u64_u32 %0, %1, %2, %3, %0\n") : "+v"(q0), "=s"(sink) : "v"(b), "v"(c));
    } else if (MODE == 3) {  // independent multiply-adds, 4 chains
      asm volatile(R4("v_mad_u64_u32 %0, %4, %5, %6, %0\n v_mad_u64_u32 %1, %4, %5, %6, %1\n v_mad_u64_u32 %2, %4, %5, %6, %2\n v_mad_u64_u32 %3, %4, %5, %6, %3\n"
                      "v_mad_u64_u32 %0, %4, %5, %6, %0\n v_mad_u64_u32 %1, %4, %5, %6, %1\n v_mad_u64_u32 %2, %4, %5, %6, %2\n v_mad_u64_u32 %3, %4, %5, %6, %3\n")
                   : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "=s"(sink) : "v"(b), "v"(c));
    } else if (MODE == 4) {  // carry chain through vcc, one s_nop between
      asm volatile(R16("v_addc_co_u32 %0, vcc, %0, %1, vcc\n s_nop 0\n v_addc_co_u32 %0, vcc, %0, %1, vcc\n s_nop 0\n") : "+v"(a0) : "v"(b) : "vcc");
    } else if (MODE == 5) {  // carry chain through vcc with an independent multiply-add between
      asm volatile(R16("v_addc_co_u32 %0, vcc, %0, %4, vcc\n v_mad_u64_u32 %1, %3, %4, %5, %1\n v_addc_co_u32 %0, vcc, %0, %4, vcc\n v_mad_u64_u32 %2, %3, %4, %5, %2\n")
                   : "+v"(a0), "+v"(q0), "+v"(q1), "=s"(sink) : "v"(b), "v"(c) : "vcc");
    } else if (MODE == 6) {  // multiply-add whose addend low word was written by the add just before (the multiplier's second chain)
      asm volatile(R16("v_add_u32 %0, %0, %3\n v_mad_u64_u32 %1, %2, %3, %4, %1\n v_add_u32 %0, %0, %3\n v_mad_u64_u32 %1, %2, %3, %4, %1\n")
                   : "+v"(a0), "+v"(q0), "=s"(sink) : "v"(b), "v"(c));
      // note: a0 and q0 are separate here; MODE 7 makes the dependency real
    } else if (MODE == 7) {  // add writes the low word of the addend pair, multiply-add reads the pair: real dependency
      asm volatile(R16("v_add_u32 v100, v101, %1\n v_mad_u64_u32 v[100:101], %0, %1, %2, v[100:101]\n v_add_u32 v100, v101, %1\n v_mad_u64_u32 v[100:101], %0, %1, %2, v[100:101]\n")
                   : "=s"(sink) : "v"(b), "v"(c) : "v100", "v101");
    } else if (MODE == 8) {  // dependent adds with one independent instruction between
      asm volatile(R16("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n") : "+v"(a0), "+v"(a1) : "v"(b));
    } else if (MODE == 9) {  // dependent multiply-adds with one independent multiply-add between (2 chains)
      asm volatile(R16("v_mad_u64_u32 %0, %2, %3, %4, %0\n v_mad_u64_u32 %1, %2, %3, %4, %1\n v_mad_u64_u32 %0, %2, %3, %4, %0\n v_mad_u64_u32 %1, %2, %3, %4, %1\n")
                   : "+v"(q0), "+v"(q1), "=s"(sink) : "v"(b), "v"(c));
    } else if (MODE == 10) {  // scalar ALU instructions only
      uint32_t s0 = seed;
      asm volatile(R16("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n") : "+s"(s0) : : "scc");
      a0 += s0;
    } else if (MODE == 11) {  // vector and scalar alternating
      uint32_t s0 = seed;
      asm volatile(R16("v_add_u32 %0, %0, %2\n s_add_u32 %1, %1, 1\n v_add_u32 %0, %0, %2\n s_add_u32 %1, %1, 1\n") : "+v"(a0), "+s"(s0) : "v"(b) : "scc");
      a0 += s0;
    }
  }
  uint32_t s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)q0 ^ (uint32_t)(q0 >> 32) ^ (uint32_t)q1 ^ (uint32_t)q2 ^ (uint32_t)q3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
int run(uint32_t* buf, const char* what, int per_iter) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int wps : {1, 2, 8}) {
    const int blocks = 256 * wps;
    k<MODE><<<blocks, 256>>>(buf, 1, 10); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); k<MODE><<<blocks, 256>>>(buf, 1, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // ns per instruction per wave (one wave per SIMD: per SIMD); at 2.4 GHz 4 cycles = 1.67 ns
    double ns = ms * 1e6 / ((double)iters * per_iter) / wps;
    printf("mode %2d %-70s waves/SIMD %d: %.3f ns per instruction per SIMD\n", MODE, what, wps, ns);
  }
  return 0;
}
int main() {
  uint32_t* buf; CK(hipMalloc(&buf, 256 * 8 * 256 * 4));
  run<0>(buf, "independent v_add_u32 (8 chains)", 32);
  run<1>(buf, "dependent v_add_u32", 32);
  run<8>(buf, "dependent v_add_u32, one independent between", 64);
  run<2>(buf, "dependent v_mad_u64_u32 (addend)", 32);
  run<9>(buf, "dependent v_mad_u64_u32, one independent between", 64);
  run<3>(buf, "independent v_mad_u64_u32 (4 chains)", 32);
  run<4>(buf, "v_addc chain through vcc + s_nop 0 (counting the adds only)", 32);
  run<5>(buf, "v_addc chain through vcc with a v_mad between (both counted)", 64);
  run<7>(buf, "v_add writes addend low word, v_mad reads it (both counted)", 64);
  run<10>(buf, "s_add_u32 dependent", 64);
  run<11>(buf, "v_add_u32 / s_add_u32 alternating (both counted)", 64);
  return 0;
}
