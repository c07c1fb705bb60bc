// BN254 scalar field on gfx950: 8 x 32-bit limbs per element, one element per lane.
//
// The reference has no field arithmetic (`Field` is a `String`,
// /root/reference/src/ast.rs:12); the operators implemented here are the ones its
// Circom backend prints (/root/reference/src/circom.rs:35-157), with circom's
// semantics over the BN254 scalar field.
//
// Values kept in witness slots are in Montgomery form (x * 2^256 mod p) and fully
// reduced to [0, p).  On gfx950 every 32-bit integer VALU instruction, the 64-bit
// multiply-add included, issues at the same rate (16 lanes/clk/SIMD), so the cost
// of an operation is its instruction count; the multiplier below is therefore
// written as a fixed instruction stream (fr_mul_asm.inc) instead of C++.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fr_mul_asm.inc"

namespace zkir {

struct Fr {
  uint32_t l[8];
};

// p, little-endian limbs
#define ZKIR_P0 0xf0000001u
#define ZKIR_P1 0x43e1f593u
#define ZKIR_P2 0x79b97091u
#define ZKIR_P3 0x2833e848u
#define ZKIR_P4 0x8181585du
#define ZKIR_P5 0xb85045b6u
#define ZKIR_P6 0xe131a029u
#define ZKIR_P7 0x30644e72u
#define ZKIR_NP0 0xefffffffu  // -p^-1 mod 2^32

__device__ __forceinline__ Fr fr_p() { return Fr{{ZKIR_P0, ZKIR_P1, ZKIR_P2, ZKIR_P3, ZKIR_P4, ZKIR_P5, ZKIR_P6, ZKIR_P7}}; }
// floor(p/2): the largest representative circom treats as non-negative
__device__ __forceinline__ Fr fr_half() {
  return Fr{{0xf8000000u, 0xa1f0fac9u, 0x3cdcb848u, 0x9419f424u, 0x40c0ac2eu, 0xdc2822dbu, 0x7098d014u, 0x18322739u}};
}
// R^2 mod p (to Montgomery form) and R mod p (Montgomery form of 1)
__device__ __forceinline__ Fr fr_r2() {
  return Fr{{0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u}};
}
__device__ __forceinline__ Fr fr_one_mont() {
  return Fr{{0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u}};
}
__device__ __forceinline__ Fr fr_zero() { return Fr{{0, 0, 0, 0, 0, 0, 0, 0}}; }
__device__ __forceinline__ Fr fr_small(uint32_t v) { return Fr{{v, 0, 0, 0, 0, 0, 0, 0}}; }

// ---- plain 256-bit helpers (not modular) ----
// r = a + b, returns carry out
__device__ __forceinline__ uint32_t u256_add(Fr& r, const Fr& a, const Fr& b) {
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = __builtin_addc(a.l[i], b.l[i], c, &c);
  return c;
}
// r = a - b, returns borrow out (1 iff a < b)
__device__ __forceinline__ uint32_t u256_sub(Fr& r, const Fr& a, const Fr& b) {
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = __builtin_subc(a.l[i], b.l[i], c, &c);
  return c;
}
__device__ __forceinline__ bool u256_lt(const Fr& a, const Fr& b) {
  Fr d;
  return u256_sub(d, a, b) != 0;
}
__device__ __forceinline__ bool u256_eq(const Fr& a, const Fr& b) {
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) x |= a.l[i] ^ b.l[i];
  return x == 0;
}
__device__ __forceinline__ bool u256_is_zero(const Fr& a) {
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) x |= a.l[i];
  return x == 0;
}
__device__ __forceinline__ Fr fr_select(bool c, const Fr& t, const Fr& f) {
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = c ? t.l[i] : f.l[i];
  return r;
}

// ---- modular add / sub / neg (any common representation) ----
__device__ __forceinline__ Fr fr_add(const Fr& a, const Fr& b) {
  Fr s, d;
  u256_add(s, a, b);  // a, b < p < 2^254: no carry out
  uint32_t bw = u256_sub(d, s, fr_p());
  return fr_select(bw != 0, s, d);
}
__device__ __forceinline__ Fr fr_sub(const Fr& a, const Fr& b) {
  Fr d, e;
  uint32_t bw = u256_sub(d, a, b);
  u256_add(e, d, fr_p());
  return fr_select(bw != 0, e, d);
}
__device__ __forceinline__ Fr fr_neg(const Fr& a) {
  Fr d;
  u256_sub(d, fr_p(), a);
  return fr_select(u256_is_zero(a), a, d);
}
// reduce x < 2^256 into [0, p) when x < 2p
__device__ __forceinline__ Fr fr_reduce_once(const Fr& x) {
  Fr d;
  uint32_t bw = u256_sub(d, x, fr_p());
  return fr_select(bw != 0, x, d);
}

// ---- Montgomery multiplication: a * b * 2^-256 mod p, inputs and result in [0, p) ----
// The streams write their results with their last eight instructions, after every input
// register has been read for the last time, so the result may share registers with an input.
__device__ __forceinline__ Fr fr_mul(const Fr& a, const Fr& b) {
  Fr r;
  uint64_t carry_sink;  // receives the multiply-adds' scalar carry-out; never read
  const uint32_t p0 = ZKIR_P0, p1 = ZKIR_P1, p2 = ZKIR_P2, p3 = ZKIR_P3, p4 = ZKIR_P4, p5 = ZKIR_P5, p6 = ZKIR_P6,
                 p7 = ZKIR_P7, np = ZKIR_NP0;
  asm volatile(ZKIR_FR_MUL_ASM
               : "=v"(r.l[0]), "=v"(r.l[1]), "=v"(r.l[2]), "=v"(r.l[3]), "=v"(r.l[4]), "=v"(r.l[5]),
                 "=v"(r.l[6]), "=v"(r.l[7]), "=&s"(carry_sink)
               : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]),
                 "v"(a.l[7]), "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]),
                 "v"(b.l[6]), "v"(b.l[7]), "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7),
                 "s"(np)
               : ZKIR_FR_MUL_CLOBBERS);
  return r;
}

// a * 2^-256 mod p for a in [0, p): leaves Montgomery form at about half the cost of a product
__device__ __forceinline__ Fr fr_redc(const Fr& a) {
  Fr r;
  uint64_t carry_sink;
  const uint32_t p0 = ZKIR_P0, p1 = ZKIR_P1, p2 = ZKIR_P2, p3 = ZKIR_P3, p4 = ZKIR_P4, p5 = ZKIR_P5, p6 = ZKIR_P6,
                 p7 = ZKIR_P7, np = ZKIR_NP0;
  asm volatile(ZKIR_FR_REDC_ASM
               : "=v"(r.l[0]), "=v"(r.l[1]), "=v"(r.l[2]), "=v"(r.l[3]), "=v"(r.l[4]), "=v"(r.l[5]),
                 "=v"(r.l[6]), "=v"(r.l[7]), "=&s"(carry_sink)
               : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]),
                 "v"(a.l[7]), "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7), "s"(np)
               : ZKIR_FR_MUL_CLOBBERS);
  return r;
}

__device__ __forceinline__ Fr fr_to_mont(const Fr& a) { return fr_mul(a, fr_r2()); }
__device__ __forceinline__ Fr fr_from_mont(const Fr& a) { return fr_redc(a); }

// ---- operators that work on the canonical integer (all arguments canonical) ----

// signed comparison of circom: x > floor(p/2) stands for x - p
// returns true iff v(a) < v(b)
__device__ __forceinline__ bool fr_signed_lt(const Fr& a, const Fr& b) {
  bool na = u256_lt(fr_half(), a);  // a negative
  bool nb = u256_lt(fr_half(), b);
  bool lt = u256_lt(a, b);
  // same sign: the raw order is the signed order; different signs: the negative one is smaller
  return (na == nb) ? lt : na;
}

// shift right by k < 256 (per lane)
__device__ __forceinline__ Fr u256_shr(const Fr& a, uint32_t k) {
  Fr r = a;
#pragma unroll
  for (int s = 2; s >= 0; s--) {  // word moves by 4, 2, 1
    const int w = 1 << s;
    bool on = (k >> (5 + s)) & 1;
    Fr t;
#pragma unroll
    for (int i = 0; i < 8; i++) t.l[i] = (i + w < 8) ? r.l[i + w] : 0u;
    r = fr_select(on, t, r);
  }
  uint32_t kb = k & 31;
  Fr o;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint32_t hi = (i + 1 < 8) ? r.l[i + 1] : 0u;
    o.l[i] = (uint32_t)((((uint64_t)hi << 32) | r.l[i]) >> kb);
  }
  return o;
}
// shift left by k < 256 (per lane), truncated to 256 bits
__device__ __forceinline__ Fr u256_shl(const Fr& a, uint32_t k) {
  Fr r = a;
#pragma unroll
  for (int s = 2; s >= 0; s--) {
    const int w = 1 << s;
    bool on = (k >> (5 + s)) & 1;
    Fr t;
#pragma unroll
    for (int i = 0; i < 8; i++) t.l[i] = (i - w >= 0) ? r.l[i - w] : 0u;
    r = fr_select(on, t, r);
  }
  uint32_t kb = k & 31;
  Fr o;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint32_t lo = (i > 0) ? r.l[i - 1] : 0u;
    o.l[i] = (uint32_t)(((((uint64_t)r.l[i] << 32) | lo) << kb) >> 32);
  }
  return o;
}

// x < 2^254 (so x < 2p): reduce into [0, p)
__device__ __forceinline__ Fr fr_mask254_reduce(Fr x) {
  x.l[7] &= 0x3fffffffu;
  return fr_reduce_once(x);
}

// `>>` and `<<` of circom; dir_right selects which one was written.
// k <= p/2: shift by k (0 when k >= 254); otherwise shift the other way by p - k.
__device__ __forceinline__ Fr fr_shift(const Fr& a, const Fr& k, bool dir_right) {
  bool big = u256_lt(fr_half(), k);
  Fr pk;
  u256_sub(pk, fr_p(), k);
  Fr amount = fr_select(big, pk, k);
  bool right = big ? !dir_right : dir_right;
  uint32_t hiw = 0;
#pragma unroll
  for (int i = 1; i < 8; i++) hiw |= amount.l[i];
  bool zero = (hiw != 0) || (amount.l[0] >= 254u);
  uint32_t s = amount.l[0] & 255u;
  Fr rs = u256_shr(a, s);
  Fr ls = fr_mask254_reduce(u256_shl(a, s));
  Fr r = fr_select(right, rs, ls);
  return fr_select(zero, fr_zero(), r);
}

// x << 1 over 256 bits, shifting `in` into bit 0
__device__ __forceinline__ void u256_shl1(Fr& x, uint32_t in) {
#pragma unroll
  for (int i = 7; i > 0; i--) x.l[i] = (x.l[i] << 1) | (x.l[i - 1] >> 31);
  x.l[0] = (x.l[0] << 1) | in;
}

// integer quotient and remainder of the unsigned representatives; a < 2^254, b != 0
__device__ __forceinline__ void u256_divrem(const Fr& a, const Fr& b, Fr& q, Fr& r) {
  // restoring division; the dividend is shifted out of the top of `aq` while the
  // quotient bits are shifted into its bottom, so no register is indexed by the loop
  Fr aq = a;
  u256_shl1(aq, 0);
  u256_shl1(aq, 0);  // bit 253 of a is now bit 255
  r = fr_zero();
  for (int it = 0; it < 254; it++) {
    uint32_t in = aq.l[7] >> 31;
    u256_shl1(r, in);
    Fr d;
    uint32_t bw = u256_sub(d, r, b);
    bool ge = (bw == 0);
    r = fr_select(ge, d, r);
    u256_shl1(aq, ge ? 1u : 0u);
  }
  q = aq;
}

// a^e, a in Montgomery form, e canonical (< 2^254); result in Montgomery form
__device__ __forceinline__ Fr fr_pow_mont(const Fr& a, const Fr& e) {
  Fr acc = fr_one_mont();
  Fr ee = e;
  u256_shl1(ee, 0);
  u256_shl1(ee, 0);
  for (int it = 0; it < 254; it++) {
    bool on = (ee.l[7] >> 31) != 0;
    u256_shl1(ee, 0);
    acc = fr_mul(acc, acc);
    Fr t = fr_mul(acc, a);
    acc = fr_select(on, t, acc);
  }
  return acc;
}
// a^(p-2) in Montgomery form (0 -> 0)
__device__ __forceinline__ Fr fr_inv_mont(const Fr& a) {
  Fr e = fr_p();
  e.l[0] -= 2u;
  return fr_pow_mont(a, e);
}

}  // namespace zkir
