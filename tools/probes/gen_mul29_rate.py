#!/usr/bin/env python3
"""Writes mul29_rate.hip: register-only loops of the 29-bit-limb field operations (tools/fr29.py), each loop one
asm statement, timed at 1, 2, 4, 5 and 8 waves per SIMD beside the 8 x 32-bit multiplier the interpreter uses today.
  kernel 0: four Montgomery products (R = 2^261) per trip, operands rotating through four register sets
  kernel 1: output conversion per trip: reduction alone, pack to 8 x 32 bits, conditional subtraction of p on the
            packed words, and the unpack back to 9 limbs (what an INPUT pays before its product by R^2)
  kernel 2: lazy addition + one parallel carry step, eight per trip
  kernel 3: exact comparison modulo p of a normal value with a lazily subtracted one: t - c + D, full carry
            propagation, quotient estimate from the top limb, q*p from a 16-entry table in LDS, limb compare
  kernel 4: pack + unpack alone, four per trip
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import fr29  # noqa: E402

V = lambda n, k=9: ["v%d" % (n + j) for j in range(k)]  # noqa: E731
X, Y, Z, W = V(10), V(20), V(30), V(40)
ACC, T = 50, "v52"
WORDS = V(54, 8)
SP = ["s%d" % (40 + j) for j in range(9)]
SNP, SINK, SCNT = "s50", "s[52:53]", "s54"


def setup(out):
    for j in range(9):
        out.append("s_mov_b32 %s, 0x%x" % (SP[j], fr29.P_LIMBS[j]))
    out.append("s_mov_b32 %s, 0x%x" % (SNP, fr29.NP))
    out.append("s_mov_b32 %s, %%[iters]" % SCNT)
    # operands: limbs from the lane's seed, top limb small (values below 2^254)
    for regs, mul in ((X, 0x9E3779B1), (Y, 0x85EBCA77)):
        for j in range(9):
            out.append("v_mul_lo_u32 %s, %%[seed], %s" % (regs[j], SP[j]))
            out.append("v_add_u32 %s, 0x%x, %s" % (regs[j], (mul * (j + 1)) & 0xFFFFFFF, regs[j]))
            out.append("v_and_b32 %s, 0x%x, %s" % (regs[j], fr29.MASK if j < 8 else 0x1FFFFF, regs[j]))


def loop(out, body):
    out.append("1:")
    out.extend(body)
    out.append("s_sub_u32 %s, %s, 1" % (SCNT, SCNT))
    out.append("s_cmp_lg_u32 %s, 0" % SCNT)
    out.append("s_cbranch_scc1 1b")


def fold(out, regs):
    out.append("v_mov_b32 %%[res], %s" % regs[0])
    for r in regs[1:]:
        out.append("v_xor_b32 %%[res], %%[res], %s" % r)


def dominating_multiple(limb_bound, top_bound):
    k = 1
    while True:
        n = fr29.limbs29(k * fr29.P)
        lend = 1
        while (lend << 29) < limb_bound + (1 << 29):
            lend += 1
        c = [n[0] + (lend << 29)] + [n[j] + (lend << 29) - lend for j in range(1, 8)] + [n[8] - lend]
        if c[8] >= top_bound and all(v >= limb_bound for v in c[:8]):
            return k, c
        k += 1


kernels = []

# 0: products
o = []
setup(o)
b = []
fr29.mul29(b, X, Y, Z, Z, ACC, SP, SNP, SINK)
fr29.mul29(b, Y, Z, W, W, ACC, SP, SNP, SINK)
fr29.mul29(b, Z, W, X, X, ACC, SP, SNP, SINK)
fr29.mul29(b, W, X, Y, Y, ACC, SP, SNP, SINK)
loop(o, b)
fold(o, X + Y)
kernels.append(("4 products, 9 x 29-bit limbs", o, 4, 0))

# 1: output conversion + unpack
o = []
setup(o)
b = []
fr29.mul29(b, X, None, Z, Z, ACC, SP, SNP, SINK, redc_only=True)
fr29.pack29(b, WORDS, Z, T)
# conditional subtraction of p on the packed words (carry chain through vcc, one instruction between)
P32 = [(fr29.P >> (32 * j)) & 0xFFFFFFFF for j in range(8)]
D32 = V(64, 8)
for j in range(8):
    b.append("v_mov_b32 %s, 0x%x" % (W[j], P32[j]))
b.append("v_sub_co_u32 %s, vcc, %s, %s" % (D32[0], WORDS[0], W[0]))
for j in range(1, 8):
    b.append("s_nop 0")
    b.append("v_subb_co_u32 %s, vcc, %s, %s, vcc" % (D32[j], WORDS[j], W[j]))
b.append("s_nop 1")
for j in range(8):
    b.append("v_cndmask_b32 %s, %s, %s, vcc" % (WORDS[j], D32[j], WORDS[j]))
fr29.unpack29(b, X, WORDS)
loop(o, b)
fold(o, X)
kernels.append(("reduction + pack + subtract p + unpack", o, 1, 0))

# 2: lazy add + parallel carry step
o = []
setup(o)
b = []
for _ in range(4):
    fr29.add29(b, X, X, Y)
    fr29.softnorm29(b, X, T)
    fr29.add29(b, Y, Y, X)
    fr29.softnorm29(b, Y, T)
loop(o, b)
fold(o, X + Y)
kernels.append(("8 x (lazy add + carry step)", o, 8, 0))

# 3: exact compare
K, DOM = dominating_multiple((1 << 29) + 8, 1 << 25)
o = []
setup(o)
# the table of q*p (normal limbs, 12 words per entry, q < 32) sits in LDS at %[lane] (an address here)
b = []
E = Z
for j in range(9):
    b.append("v_add_u32 %s, 0x%x, %s" % (E[j], DOM[j], X[j]))
    b.append("v_sub_u32 %s, %s, %s" % (E[j], E[j], Y[j]))
fr29.fullnorm29(b, E, T)
b.append("v_mul_hi_u32 %s, %s, %s" % (T, E[8], "s55"))       # q
b.append("v_mul_u32_u24 %s, 48, %s" % (T, T))               # table offset
b.append("v_add_u32 %s, %s, %%[lane]" % (T, T))
b.append("ds_read_b128 v[64:67], %s" % T)
b.append("ds_read_b128 v[68:71], %s offset:16" % T)
b.append("ds_read_b32 v72, %s offset:32" % T)
b.append("s_waitcnt lgkmcnt(0)")
for j in range(9):
    b.append("v_xor_b32 %s, %s, v%d" % (E[j], E[j], 64 + j))
for j in range(1, 9):
    b.append("v_or_b32 %s, %s, %s" % (E[0], E[0], E[j]))
b.append("v_cmp_ne_u32 vcc, 0, %s" % E[0])
b.append("s_nop 1")
b.append("v_cndmask_b32 %s, 0, 1, vcc" % T)
b.append("v_add_u32 %s, %s, %s" % (Y[0], Y[0], T))  # keeps the result alive and the operands changing
b.append("v_and_b32 %s, 0x%x, %s" % (Y[0], fr29.MASK, Y[0]))
o.append("s_mov_b32 s55, 1355")
loop(o, b)
fold(o, X + Y)
kernels.append(("exact compare modulo p (k = %d)" % K, o, 1, 1))

# 4: pack + unpack
o = []
setup(o)
b = []
for _ in range(4):
    fr29.pack29(b, WORDS, X, T)
    fr29.unpack29(b, X, WORDS)
loop(o, b)
fold(o, X)
kernels.append(("4 x (pack + unpack)", o, 4, 0))

clob = ", ".join('"v%d"' % k for k in range(10, 80)) + ", " + ", ".join('"s%d"' % k for k in range(40, 56)) + ', "vcc", "scc", "memory"'
src = ['// generated by gen_mul29_rate.py -- do not edit',
       '//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../zkir_amd/csrc/hip -o mul29_rate mul29_rate.hip',
       '#include <hip/hip_runtime.h>', '#include <stdint.h>', '#include <stdio.h>', '#include "fr32.cuh"',
       '#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP error %s line %d\\n", hipGetErrorString(e_), __LINE__); return 1;}}while(0)',
       'using namespace zkir;']
for n, (name, lines, per, lds) in enumerate(kernels):
    src.append("// %s" % name)
    src.append("__global__ void __launch_bounds__(256) k%d(uint32_t* out, uint32_t seed_in, int iters) {" % n)
    if lds:
        src.append("  __shared__ uint32_t table[32 * 12];")
        src.append("  for (int q = threadIdx.x; q < 32; q += 256) {")
        src.append("    uint64_t carry = 0; const uint32_t pl[9] = {%s};" % ", ".join("0x%xu" % v for v in fr29.P_LIMBS))
        src.append("    for (int j = 0; j < 9; j++) { uint64_t t = (uint64_t)pl[j] * q + carry; table[q * 12 + j] = j < 8 ? (uint32_t)(t & 0x1fffffff) : (uint32_t)t; carry = t >> 29; }")
        src.append("    table[q * 12 + 9] = table[q * 12 + 10] = table[q * 12 + 11] = 0; }")
        src.append("  __syncthreads();")
    src.append("  uint32_t seed = seed_in + threadIdx.x * 2654435761u + blockIdx.x, lane = %s, res;" % ("(uint32_t)(uintptr_t)table" if lds else "threadIdx.x"))
    src.append("  asm volatile(")
    for l in lines:
        src.append('    "%s\\n"' % l)
    src.append('    : [res] "=v"(res) : [seed] "v"(seed), [lane] "v"(lane), [iters] "s"(iters) : %s);' % clob)
    if lds:
        src.append("  res ^= table[threadIdx.x % (32 * 12)];")
    src.append("  out[blockIdx.x * blockDim.x + threadIdx.x] = res;")
    src.append("}")
src.append(r'''
// the 8 x 32-bit multiplier of the interpreter, same shape as kernel 0 (two products per trip)
__global__ void __launch_bounds__(256) k_ref(uint32_t* out, uint32_t seed, int iters) {
  Fr x = fr_one_mont(), y = fr_r2();
  x.l[0] ^= threadIdx.x + seed; x.l[7] &= 0x0fffffffu; y.l[0] ^= blockIdx.x; y.l[7] &= 0x0fffffffu;
  for (int it = 0; it < iters; it++) { x = fr_mul(x, y); y = fr_mul(y, x); }
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s ^= x.l[i] ^ y.l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
int run(K kern, const char* name, int per_trip, int iters, uint32_t* buf, int n_cu) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-44s", name);
  for (int wps : {1, 2, 4, 5, 8}) {
    const int blocks = n_cu * wps;
    kern<<<blocks, 256>>>(buf, 1, 8); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) kern<<<blocks, 256>>>(buf, 1, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double per_s = (double)blocks * 256 * per_trip * iters * 3 / (ms * 1e-3);
    const double cyc = ms * 1e6 / 3.0 / ((double)iters * per_trip) / wps * 2.4;  // per wave per item per SIMD
    printf("  w%d: %7.2f G/s %6.0f cyc", wps, per_s / 1e9, cyc);
  }
  printf("\n");
  return 0;
}
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  printf("%s CUs=%d; items per second chip-wide, and cycles per item per wave per SIMD at 2.4 GHz\n", prop.gcnArchName, n_cu);
  uint32_t* buf; CK(hipMalloc(&buf, (size_t)n_cu * 8 * 256 * 4));
  for (int r = 0; r < 10; r++) k_ref<<<n_cu * 8, 256>>>(buf, 1, 2000);  // warm the clock
  CK(hipDeviceSynchronize());
  if (run(k_ref, "product, 8 x 32-bit limbs (today)", 2, 2000, buf, n_cu)) return 1;''')
for n, (name, lines, per, lds) in enumerate(kernels):
    src.append('  if (run(k%d, "%s", %d, %d, buf, n_cu)) return 1;' % (n, name, per, 1000 if per > 1 else 3000))
src.append("  return 0;\n}")
with open(os.path.join(HERE, "mul29_rate.hip"), "w") as f:
    f.write("\n".join(src) + "\n")
print("wrote mul29_rate.hip:", [(name, len(lines)) for name, lines, _, _ in kernels])
