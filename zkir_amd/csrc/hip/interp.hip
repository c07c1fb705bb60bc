// Tape interpreter kernels for gfx950: one lane per input vector, the tape is
// wave-uniform (fetched with scalar loads), witness slots live in LDS (the
// on-chip window) or in a per-workgroup region of device memory.
//
// The reference has no evaluator to mirror (/root/reference/src/ast.rs:12 makes
// `Field` a `String`); what a tape op means is the Circom operator its Expr
// variant prints as (/root/reference/src/circom.rs:35-157).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include "../../../include/zkir_hip.h"
#include "device_structs.h"
#include "fr32.cuh"

namespace zkir {

constexpr int kBlock = ZKIR_BLOCK_THREADS;  // lanes (input vectors) per workgroup pass

// Tapes, constants and tape descriptors are read-only for the whole launch and
// addressed wave-uniformly.  Reading them through the constant address space makes
// the compiler fetch them with scalar loads into SGPRs, so the opcode dispatch is
// scalar branches, not per-lane exec masking.
#if defined(__HIP_DEVICE_COMPILE__)
#define ZKIR_UNIFORM __attribute__((address_space(4)))
#else
#define ZKIR_UNIFORM  // host pass only parses the kernels
#endif
typedef const ZKIR_UNIFORM zkir_op* uniform_ops_t;
typedef const ZKIR_UNIFORM uint32_t* uniform_u32_t;
typedef const ZKIR_UNIFORM TapeDesc* uniform_desc_t;

// ---- slot storage -------------------------------------------------------
// A slot of a 256-lane workgroup is two planes of 256 x 16 bytes: lane t keeps
// limbs 0..3 at plane 0, limbs 4..7 at plane 1.  Consecutive lanes are 16 bytes
// apart, so every access is one 128-bit instruction per plane: conflict-free
// in LDS, fully coalesced in device memory.
// The LDS window is addressed through the array itself, never through a pointer kept in a
// struct: a pointer would decay to the generic address space and every access would become a
// flat_load / flat_store instead of ds_read_b128 / ds_write_b128.
extern __shared__ uint4 zkir_lds_slots[];
__device__ __forceinline__ Fr load_lds_slot(uint32_t slot) {
  const uint32_t i = slot * 2 * kBlock + threadIdx.x;
  const uint4 lo = zkir_lds_slots[i], hi = zkir_lds_slots[i + kBlock];
  return Fr{{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
}
__device__ __forceinline__ void store_lds_slot(uint32_t slot, const Fr& v) {
  const uint32_t i = slot * 2 * kBlock + threadIdx.x;
  zkir_lds_slots[i] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  zkir_lds_slots[i + kBlock] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

__device__ __forceinline__ Fr load_slot(const uint4* base, uint32_t slot) {
  const uint4* p = base + (size_t)slot * 2 * kBlock + threadIdx.x;
  uint4 lo = p[0], hi = p[kBlock];
  return Fr{{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
}
__device__ __forceinline__ void store_slot(uint4* base, uint32_t slot, const Fr& v) {
  uint4* p = base + (size_t)slot * 2 * kBlock + threadIdx.x;
  p[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  p[kBlock] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

struct LaneCtx {
  uint4* gslots;         // this workgroup's device-memory slots
  uniform_u32_t consts;  // Montgomery-form constant pool of the tape
};

__device__ __forceinline__ Fr load_operand(uint32_t o, const Fr& acc, const LaneCtx& cx) {
  const uint32_t kind = o >> ZKIR_KIND_SHIFT, idx = o & ZKIR_INDEX_MASK;
  Fr r;
  switch (kind) {
    case ZKIR_KIND_ACC:
      r = acc;
      break;
    case ZKIR_KIND_LSLOT:
      r = load_lds_slot(idx);
      break;
    case ZKIR_KIND_CONST: {
      uniform_u32_t c = cx.consts + (size_t)idx * 8;
      r = Fr{{c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]}};
      break;
    }
    default:
      r = load_slot(cx.gslots, idx);
      break;
  }
  return r;
}
__device__ __forceinline__ void store_operand(uint32_t o, const Fr& v, const LaneCtx& cx) {
  const uint32_t kind = o >> ZKIR_KIND_SHIFT, idx = o & ZKIR_INDEX_MASK;
  switch (kind) {
    case ZKIR_KIND_LSLOT:
      store_lds_slot(idx, v);
      break;
    case ZKIR_KIND_GSLOT:
      store_slot(cx.gslots, idx, v);
      break;
    default:
      break;
  }
}

// ---- the operators that are not + - * -----------------------------------
// Kept out of line: they are rare in tapes and each holds several copies of the
// multiplier.  x, y are Montgomery form; the result is Montgomery form.
struct SlowResult {
  Fr r;
  uint32_t fault;  // 1 when the lane divided by zero
};
__device__ __forceinline__ Fr slow_op_body(uint32_t code, const Fr& x, const Fr& y, uint32_t* fault) {
  const Fr one = fr_one_mont();
  switch (code) {
    case ZKIR_OP_POW:
      return fr_pow_mont(x, fr_from_mont(y));
    case ZKIR_OP_DIV: {
      if (u256_is_zero(y)) *fault += 1;
      return fr_mul(x, fr_inv_mont(y));  // inverse of 0 is 0
    }
    default:
      break;
  }
  // everything below works on the canonical integers
  const Fr a = fr_from_mont(x);
  const Fr b = fr_from_mont(y);
  bool flag = false, is_flag = true;
  Fr r = fr_zero();
  switch (code) {
    case ZKIR_OP_LT: flag = fr_signed_lt(a, b); break;
    case ZKIR_OP_LE: flag = !fr_signed_lt(b, a); break;
    case ZKIR_OP_GT: flag = fr_signed_lt(b, a); break;
    case ZKIR_OP_GE: flag = !fr_signed_lt(a, b); break;
    case ZKIR_OP_EQ: flag = u256_eq(a, b); break;
    case ZKIR_OP_AND: flag = !u256_is_zero(a) && !u256_is_zero(b); break;
    case ZKIR_OP_OR: flag = !u256_is_zero(a) || !u256_is_zero(b); break;
    default: is_flag = false; break;
  }
  if (is_flag) return fr_select(flag, one, r);
  switch (code) {
    case ZKIR_OP_IDIV:
    case ZKIR_OP_REM: {
      const bool z = u256_is_zero(b);
      if (z) *fault += 1;
      Fr q, m;
      u256_divrem(a, fr_select(z, fr_small(1), b), q, m);
      r = (code == ZKIR_OP_IDIV) ? q : m;
      r = fr_select(z, fr_zero(), r);
      break;
    }
    case ZKIR_OP_BAND:
#pragma unroll
      for (int i = 0; i < 8; i++) r.l[i] = a.l[i] & b.l[i];
      break;  // <= min(a, b) < p
    case ZKIR_OP_BOR:
#pragma unroll
      for (int i = 0; i < 8; i++) r.l[i] = a.l[i] | b.l[i];
      r = fr_reduce_once(r);  // < 2^254 < 2p
      break;
    case ZKIR_OP_BXOR:
#pragma unroll
      for (int i = 0; i < 8; i++) r.l[i] = a.l[i] ^ b.l[i];
      r = fr_reduce_once(r);
      break;
    case ZKIR_OP_BNOT:
#pragma unroll
      for (int i = 0; i < 8; i++) r.l[i] = ~a.l[i];
      r = fr_mask254_reduce(r);
      break;
    case ZKIR_OP_BNOT256: {
      // 2^256 - 1 - a is below 6p: subtract 4p, 2p, p as needed
#pragma unroll
      for (int i = 0; i < 8; i++) r.l[i] = ~a.l[i];
      Fr p2, p4, d;
      u256_add(p2, fr_p(), fr_p());
      u256_add(p4, p2, p2);
      uint32_t bw = u256_sub(d, r, p4);
      r = fr_select(bw != 0, r, d);
      bw = u256_sub(d, r, p2);
      r = fr_select(bw != 0, r, d);
      r = fr_reduce_once(r);
      break;
    }
    case ZKIR_OP_SHR: r = fr_shift(a, b, true); break;
    case ZKIR_OP_SHL: r = fr_shift(a, b, false); break;
    default: break;
  }
  return fr_to_mont(r);
}
__device__ __noinline__ SlowResult slow_op(uint32_t code, Fr x, Fr y) {
  SlowResult o;
  o.fault = 0;
  o.r = slow_op_body(code, x, y, &o.fault);
  return o;
}

// ---- the interpreter ------------------------------------------------------
// Work item w = (tape, pass).  Workgroups draw items from one device-wide counter (zeroed
// by the host before the launch) until it runs past the last item, so the grid can be sized
// to the machine, the per-workgroup slot region is reused from item to item, and a
// workgroup that starts late (grid larger than what is resident) just finds less work left.
// kFull = false is the variant for tapes made only of + - * (and data movement): it
// carries no call to the out-of-line operators, which keeps its register count down.
template <bool kFull>
__global__ void __launch_bounds__(kBlock, 2)
zkir_interp_kernel(const TapeDesc* __restrict__ tapes, uint32_t n_tapes, uint32_t passes_per_tape,
                   uint64_t n_vectors,  // vectors per tape
                   const uint32_t* __restrict__ inputs, uint32_t* __restrict__ outputs,
                   uint4* __restrict__ gslot_pool, uint32_t gslots_per_block,
                   uint64_t* __restrict__ unsat_per_tape, uint64_t* __restrict__ stats,
                   uint32_t* __restrict__ work_counter, uint64_t* __restrict__ faults_per_tape) {
  __shared__ uint32_t next_item;
  LaneCtx cx;
  cx.gslots = gslot_pool + (size_t)blockIdx.x * gslots_per_block * 2 * kBlock;

  const uint64_t total = (uint64_t)n_tapes * passes_per_tape;  // the host keeps this below 2^32
  for (;;) {
    if (threadIdx.x == 0) next_item = atomicAdd(work_counter, 1u);
    __syncthreads();
    const uint64_t w = next_item;
    __syncthreads();  // everyone has read it before lane 0 draws again
    if (w >= total) break;
    const uint32_t tape_i = (uint32_t)(w / passes_per_tape);
    const uint32_t pass = (uint32_t)(w % passes_per_tape);
    const TapeDesc td = ((uniform_desc_t)tapes)[tape_i];
    cx.consts = (uniform_u32_t)td.consts;
    uniform_ops_t ops = (uniform_ops_t)td.ops;
    const uint64_t vec = (uint64_t)pass * kBlock + threadIdx.x;
    const bool live = vec < n_vectors;
    const uint64_t v = live ? vec : 0;  // idle lanes recompute vector 0 and report nothing
    const uint32_t* in = inputs + td.in_word_off + v;
    uint32_t* out = outputs ? outputs + td.out_word_off + v : nullptr;

    Fr acc = fr_zero();
    uint32_t fault = 0, first_bad = 0xffffffffu, check_k = 0;
    uint64_t wave_unsat = 0;

    // Every value-defining op leaves its result in `acc` (and stores it if the tape asks);
    // CHECK and OUTPUT leave `acc` alone.  The next op is fetched (scalar load) while the
    // current one runs.  An operand of kind ACC is used in place: the multiplier and the
    // adder read `acc` directly and may write their result over it.
    zkir_op next_op = ops[0];
    for (uint32_t pc = 0; pc < td.n_ops; pc++) {
      const zkir_op op = next_op;
      if (pc + 1 < td.n_ops) next_op = ops[pc + 1];
      const uint32_t code = op.code & 0xffu;
      const bool a_acc = (op.a >> ZKIR_KIND_SHIFT) == ZKIR_KIND_ACC;
      const bool b_acc = (op.b >> ZKIR_KIND_SHIFT) == ZKIR_KIND_ACC;

      if (code == ZKIR_OP_MUL) {
        if (a_acc) {
          const Fr y = load_operand(op.b, acc, cx);
          acc = fr_mul(acc, y);
        } else {
          const Fr x = load_operand(op.a, acc, cx);
          const Fr y = load_operand(op.b, acc, cx);
          acc = fr_mul(x, y);
        }
        store_operand(op.dst, acc, cx);
      } else if (code == ZKIR_OP_ADD) {
        if (a_acc) {
          const Fr y = load_operand(op.b, acc, cx);
          acc = fr_add(acc, y);
        } else {
          const Fr x = load_operand(op.a, acc, cx);
          const Fr y = load_operand(op.b, acc, cx);
          acc = fr_add(x, y);
        }
        store_operand(op.dst, acc, cx);
      } else if (code == ZKIR_OP_SUB) {
        if (a_acc) {
          const Fr y = load_operand(op.b, acc, cx);
          acc = fr_sub(acc, y);
        } else if (b_acc) {
          const Fr x = load_operand(op.a, acc, cx);
          acc = fr_sub(x, acc);
        } else {
          const Fr x = load_operand(op.a, acc, cx);
          const Fr y = load_operand(op.b, acc, cx);
          acc = fr_sub(x, y);
        }
        store_operand(op.dst, acc, cx);
      } else if (code == ZKIR_OP_CHECK) {
        Fr t;
        if (op.b == td.one_word) {  // value * 1 == signal: the left factor as it is (x * 1 is x bit for bit)
          t = load_operand(op.a, acc, cx);
        } else if (a_acc) {
          const Fr y = load_operand(op.b, acc, cx);
          t = fr_mul(acc, y);
        } else {
          const Fr x = load_operand(op.a, acc, cx);
          const Fr y = load_operand(op.b, acc, cx);
          t = fr_mul(x, y);
        }
        const Fr c = load_operand(op.dst, acc, cx);
        const bool bad = live && !u256_eq(t, c);
        wave_unsat += __popcll(__ballot(bad));
        // the constraint's number: carried by the op (number + 1 in bits 8..31), else its position among the CHECK ops
        const uint32_t id = (op.code >> 8) ? (op.code >> 8) - 1u : check_k;
        if (bad && id < first_bad) first_bad = id;
        check_k++;
      } else if (code == ZKIR_OP_OUTPUT || code == ZKIR_OP_OUTPUT_SCALED) {
        if (out) {  // nobody wants the outputs: skip the conversion too
          const Fr x = load_operand(op.a, acc, cx);
          // a tape value x is held as x * 2^256 mod p: the output times 2^256 is the held words as they are
          const Fr r = code == ZKIR_OP_OUTPUT ? fr_redc(x) : x;
          if (live) {
            uint32_t* q = out + (size_t)op.dst * 8 * n_vectors;
#pragma unroll
            for (int i = 0; i < 8; i++) q[(size_t)i * n_vectors] = r.l[i];
          }
        }
      } else if (code == ZKIR_OP_INPUT || code == ZKIR_OP_INPUT_SCALED) {
        const uint32_t* q = in + (size_t)(op.a & ZKIR_INDEX_MASK) * 8 * n_vectors;
        Fr x;
#pragma unroll
        for (int i = 0; i < 8; i++) x.l[i] = q[(size_t)i * n_vectors];
        acc = code == ZKIR_OP_INPUT ? fr_mul(x, fr_r2()) : x;  // (the input times 2^-256 is held as the input's own words)
        store_operand(op.dst, acc, cx);
      } else if (code == ZKIR_OP_COPY) {
        acc = load_operand(op.a, acc, cx);
        store_operand(op.dst, acc, cx);
      } else if (code == ZKIR_OP_NEG) {
        acc = fr_neg(load_operand(op.a, acc, cx));
        store_operand(op.dst, acc, cx);
      } else if (kFull && code != ZKIR_OP_NOP) {
        const Fr x = load_operand(op.a, acc, cx);
        const Fr y = code >= ZKIR_OP_ADD ? load_operand(op.b, acc, cx) : fr_zero();
        const SlowResult sr = slow_op(code, x, y);
        acc = sr.r;
        fault |= sr.fault;
        store_operand(op.dst, acc, cx);
      }
    }

    // one atomic per wave for the counters, one per failing lane for the minimum
    const uint64_t wave_faults = __popcll(__ballot(live && fault != 0));
    if ((threadIdx.x & 63) == 0) {
      if (wave_unsat) atomicAdd((unsigned long long*)&unsat_per_tape[td.unsat_index], (unsigned long long)wave_unsat);
      if (wave_faults) {
        atomicAdd((unsigned long long*)&stats[1], (unsigned long long)wave_faults);
        if (faults_per_tape) atomicAdd((unsigned long long*)&faults_per_tape[td.unsat_index], (unsigned long long)wave_faults);
      }
    }
    if (first_bad != 0xffffffffu)
      atomicMin((unsigned long long*)&stats[0], (unsigned long long)((vec << 32) | first_bad));
  }
}

// ---- the lean interpreter ---------------------------------------------------
// Same work-item loop, same results, for tapes made of + - * and data movement; the per-op
// loop is one generated instruction stream (tools/gen_interp_asm.py) instead of compiled
// C++: operands land directly in the registers the multiplier reads, an accumulator operand
// costs nothing, and a wait on vector memory is issued only by an op that loaded from it.
// Every access the stream makes goes through a buffer descriptor carrying the byte count of
// the region (tape, constants, this workgroup's slots, one input or output element), or to
// LDS.  Slot layout differs from the C++ kernels (which is fine: slots do not outlive a work
// item): slot s, wave w, plane h, lane l at s*8192 + w*2048 + h*1024 + l*16.
// Limits the host checks before choosing this kernel: 32 * n_vectors < 2^32.
#include "interp_core_asm.inc"

// strands != 0 (4 or 8): every tape of the launch is a strand tape of that many strands.  The
// wavefronts of a workgroup (64 x strands threads) then run the op streams of the tape on the
// SAME 64 vectors (a pass is 64 vectors, not 256), each with its own rows of the slots, and meet
// in the exchange slots that follow the window slots in LDS.  A slot row holds four wavefronts,
// so a workgroup of eight keeps two sets of rows, in LDS and in device memory: wavefronts 4..7
// use the second set (`gslots_per_block` counts both).  The tape's barriers pair up across the
// wavefronts because the host only uploads strand tapes whose strands hold the same number of them.
__global__ void __launch_bounds__(2 * kBlock)
zkir_interp_lean_kernel(const TapeDesc* __restrict__ tapes, uint32_t n_tapes, uint32_t passes_per_tape,
                        uint64_t n_vectors, const uint32_t* __restrict__ inputs, uint32_t* __restrict__ outputs,
                        uint4* __restrict__ gslot_pool, uint32_t gslots_per_block,
                        uint64_t* __restrict__ unsat_per_tape, uint64_t* __restrict__ stats,
                        uint32_t* __restrict__ work_counter, uint32_t victim_row, uint32_t strands, uint32_t lslots, uint32_t turns) {
  __shared__ uint32_t next_item;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
  const uint32_t row_sets = strands == 8u ? 2u : 1u, set = wave >> 2;  // which set of slot rows this wavefront uses
  const uint32_t gslots_per_set = gslots_per_block / row_sets;
  const uint64_t gslots = (uint64_t)(uintptr_t)(gslot_pool + ((size_t)blockIdx.x * gslots_per_block + (size_t)set * gslots_per_set) * 2 * kBlock);
  const uint32_t gslot_bytes = gslots_per_set * (uint32_t)(2 * kBlock * sizeof(uint4));
  // the low half of a generic pointer into LDS is the LDS byte address; with a victim row the
  // first 8 KiB of the dynamic allocation are that row and the window slots follow
  const uint32_t lds_rows = (uint32_t)(uintptr_t)zkir_lds_slots + (victim_row ? 8192u : 0u);
  // (wave-uniform values an asm statement takes in scalar registers: said so explicitly, or the compiler may keep them in vector ones)
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_rows + set * lslots * 8192u);
  const uint32_t xbase = __builtin_amdgcn_readfirstlane(lds_rows + row_sets * lslots * 8192u);
  const uint32_t evict_below = victim_row ? 0x20000000u : 0u;
  const uint32_t v_lo = (wave & 3u) * 2048u + lane * 16u, v_x = lane * 16u;
  const uint32_t n4 = (uint32_t)n_vectors * 4u;
  // priority turns of a strand tape (turns = CUs | log2 of the turn in clock ticks << 16; 0: none): workgroups are
  // placed CU by CU in the order of their numbers, so the ones that share a CU are one CU count apart and the
  // quotient tells them apart
  const uint32_t rot = turns ? __builtin_amdgcn_readfirstlane(blockIdx.x / (turns & 0xffffu)) : 0xffffffffu;
  const uint32_t rot_shift = turns >> 16;

  const uint64_t total = (uint64_t)n_tapes * passes_per_tape;
  for (;;) {
    if (threadIdx.x == 0) next_item = atomicAdd(work_counter, 1u);
    __syncthreads();
    const uint64_t w = next_item;
    __syncthreads();
    if (w >= total) break;
    const uint32_t tape_i = (uint32_t)(w / passes_per_tape);
    const uint32_t pass = (uint32_t)(w % passes_per_tape);
    uniform_desc_t tdp = (uniform_desc_t)tapes + tape_i;
    const TapeDesc td = *tdp;
    const uint64_t vec = strands ? (uint64_t)pass * 64u + lane : (uint64_t)pass * kBlock + threadIdx.x;
    const bool live = vec < n_vectors;
    const uint32_t v_vec4 = live ? (uint32_t)vec * 4u : 0u;  // idle lanes recompute vector 0 and report nothing
    const uint64_t live_mask = __ballot(live);
    // indexed through the pointer, not the copy: a run-time index into `td` would move it to scratch
    const uint32_t first_op = strands ? tdp->strand_begin[wave] : 0u;
    const uint32_t n_ops = strands ? tdp->strand_ops[wave] : td.n_ops;
    const uint64_t ops = (uint64_t)(uintptr_t)(td.ops + first_op), consts = (uint64_t)(uintptr_t)td.consts;
    const uint64_t in_base = (uint64_t)(uintptr_t)(inputs + td.in_word_off);
    const uint64_t out_base = outputs ? (uint64_t)(uintptr_t)(outputs + td.out_word_off) : 0;  // 0: not wanted

    uint64_t wave_unsat;
    uint32_t first_bad;
    asm volatile(ZKIR_INTERP_CORE_ASM
                 : [unsat] "=s"(wave_unsat), [first] "=v"(first_bad)
                 : [ops] "s"(ops), [n_ops] "s"(n_ops), [consts] "s"(consts), [n_consts] "s"(td.n_consts),
                   [gslots] "s"(gslots), [gslot_bytes] "s"(gslot_bytes), [lds_base] "s"(lds_base),
                   [in_base] "s"(in_base), [out_base] "s"(out_base), [n4] "s"(n4), [live] "s"(live_mask),
                   [evict_below] "s"(evict_below), [xbase] "s"(xbase), [one_word] "s"(td.one_word), [rot] "s"(rot), [rot_shift] "s"(rot_shift), [v_lo] "v"(v_lo), [v_vec4] "v"(v_vec4), [v_x] "v"(v_x)
                 : ZKIR_INTERP_CORE_CLOBBERS);

    if (lane == 0 && wave_unsat)
      atomicAdd((unsigned long long*)&unsat_per_tape[td.unsat_index], (unsigned long long)wave_unsat);
    if (live && first_bad != 0xffffffffu)
      atomicMin((unsigned long long*)&stats[0], (unsigned long long)((vec << 32) | first_bad));
  }
}

// ---- the same interpreter on 9 limbs of 29 bits --------------------------------
// The stream of tools/gen_interp29_asm.py: values are 9 limbs of 29 bits in Montgomery form with R = 2^261, so a
// product is 162 multiply-adds into 64-bit column sums with no carry-writing add (on gfx950 those cost as much
// as a multiply); sums and differences are limb-wise and reduced through a table of the first 32 multiples of p
// that the workgroup keeps at the start of its LDS.  Same work-item loop, same results.
// Slot layout: slot s, wave w at s*9216 + w*2304; plane 0 (limbs 0..3) lane*16, plane 1 (limbs 4..7) 1024 + lane*16,
// plane 2 (limb 8) 2048 + lane*4.  Constants come from the tape's second pool (12 words each, this form).
// Limits the host checks before choosing this kernel: 32 * n_vectors < 2^32 and 9216 * gslots_per_block < 2^32.
#include "interp29_core_asm.inc"

constexpr uint32_t kSlot29 = 9216, kWave29 = 2304, kTable29 = 2048;
__constant__ uint32_t zkir_p29[9] = {ZKIR_FR29_P_LIMBS};

__global__ void __launch_bounds__(2 * kBlock)
zkir_interp_lean29_kernel(const TapeDesc* __restrict__ tapes, uint32_t n_tapes, uint32_t passes_per_tape,
                          uint64_t n_vectors, const uint32_t* __restrict__ inputs, uint32_t* __restrict__ outputs,
                          uint4* __restrict__ gslot_pool, uint32_t gslots_per_block,
                          uint64_t* __restrict__ unsat_per_tape, uint64_t* __restrict__ stats,
                          uint32_t* __restrict__ work_counter, uint32_t victim_row, uint32_t strands, uint32_t lslots, uint32_t turns) {
  __shared__ uint32_t next_item;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
  const uint32_t row_sets = strands == 8u ? 2u : 1u, set = wave >> 2;  // as in zkir_interp_lean_kernel
  const uint32_t gslots_per_set = gslots_per_block / row_sets;
  const uint64_t gslots = (uint64_t)(uintptr_t)((const char*)gslot_pool + ((size_t)blockIdx.x * gslots_per_block + (size_t)set * gslots_per_set) * kSlot29);
  const uint32_t gslot_bytes = gslots_per_set * kSlot29;
  // LDS: the table of multiples of p, then the victim row (if any), the window slots, the exchange slots
  const uint32_t lds_tab = (uint32_t)(uintptr_t)zkir_lds_slots;
  const uint32_t lds_rows = lds_tab + kTable29 + (victim_row ? kSlot29 : 0u);
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_rows + set * lslots * kSlot29);
  const uint32_t xbase = __builtin_amdgcn_readfirstlane(lds_rows + row_sets * lslots * kSlot29);
  const uint32_t evict_below = victim_row ? 0x20000000u : 0u;
  const uint32_t v_lo = (wave & 3u) * kWave29 + lane * 16u, v_lo4 = (wave & 3u) * kWave29 + 2048u + lane * 4u;
  const uint32_t v_x = lane * 16u, v_x4 = 2048u + lane * 4u;
  const uint32_t n4 = (uint32_t)n_vectors * 4u;
  const uint32_t rot = turns ? __builtin_amdgcn_readfirstlane(blockIdx.x / (turns & 0xffffu)) : 0xffffffffu;  // as in zkir_interp_lean_kernel
  const uint32_t rot_shift = turns >> 16;
  if (threadIdx.x < 32) {  // entry q: the limbs of q*p, 12 words (9 used)
    uint32_t* row = (uint32_t*)zkir_lds_slots + threadIdx.x * 12;
    uint64_t carry = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      const uint64_t t = (uint64_t)zkir_p29[j] * threadIdx.x + carry;
      row[j] = j < 8 ? (uint32_t)(t & 0x1fffffffu) : (uint32_t)t;
      carry = t >> 29;
    }
    row[9] = row[10] = row[11] = 0;
  }
  __syncthreads();

  const uint64_t total = (uint64_t)n_tapes * passes_per_tape;
  for (;;) {
    if (threadIdx.x == 0) next_item = atomicAdd(work_counter, 1u);
    __syncthreads();
    const uint64_t w = next_item;
    __syncthreads();
    if (w >= total) break;
    const uint32_t tape_i = (uint32_t)(w / passes_per_tape);
    const uint32_t pass = (uint32_t)(w % passes_per_tape);
    uniform_desc_t tdp = (uniform_desc_t)tapes + tape_i;
    const TapeDesc td = *tdp;
    const uint64_t vec = strands ? (uint64_t)pass * 64u + lane : (uint64_t)pass * kBlock + threadIdx.x;
    const bool live = vec < n_vectors;
    const uint32_t v_vec4 = live ? (uint32_t)vec * 4u : 0u;  // idle lanes recompute vector 0 and report nothing
    const uint64_t live_mask = __ballot(live);
    const uint32_t first_op = strands ? tdp->strand_begin[wave] : 0u;
    const uint32_t n_ops = strands ? tdp->strand_ops[wave] : td.n_ops;
    const uint64_t ops = (uint64_t)(uintptr_t)(td.ops + first_op), consts = (uint64_t)(uintptr_t)td.consts29;
    const uint64_t in_base = (uint64_t)(uintptr_t)(inputs + td.in_word_off);
    const uint64_t out_base = outputs ? (uint64_t)(uintptr_t)(outputs + td.out_word_off) : 0;  // 0: not wanted

    uint64_t wave_unsat;
    uint32_t first_bad;
    asm volatile(ZKIR_INTERP29_CORE_ASM
                 : [unsat] "=s"(wave_unsat), [first] "=v"(first_bad)
                 : [ops] "s"(ops), [n_ops] "s"(n_ops), [consts] "s"(consts), [n_consts] "s"(td.n_consts),
                   [gslots] "s"(gslots), [gslot_bytes] "s"(gslot_bytes), [lds_base] "s"(lds_base), [lds_tab] "s"(lds_tab),
                   [in_base] "s"(in_base), [out_base] "s"(out_base), [n4] "s"(n4), [live] "s"(live_mask),
                   [evict_below] "s"(evict_below), [xbase] "s"(xbase), [one_word] "s"(td.one_word), [rot] "s"(rot), [rot_shift] "s"(rot_shift), [v_lo] "v"(v_lo), [v_lo4] "v"(v_lo4),
                   [v_vec4] "v"(v_vec4), [v_x] "v"(v_x), [v_x4] "v"(v_x4)
                 : ZKIR_INTERP29_CORE_CLOBBERS);

    if (lane == 0 && wave_unsat)
      atomicAdd((unsigned long long*)&unsat_per_tape[td.unsat_index], (unsigned long long)wave_unsat);
    if (live && first_bad != 0xffffffffu)
      atomicMin((unsigned long long*)&stats[0], (unsigned long long)((vec << 32) | first_bad));
  }
}

// constants of a tape in the form the 29-bit stream reads: canonical 8 x 32-bit words in, 12 words out
// (9 limbs of 29 bits of c * 2^261 mod p, then 3 words of padding); src is not modified
__global__ void zkir_consts_to_mont29_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t a[9];
  const uint32_t r2[9] = {ZKIR_FR29_R2_LIMBS};
  {  // unpack
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = src[(size_t)i * 8 + k];
#pragma unroll
    for (int j = 0; j < 9; j++) {
      const int bit = 29 * j, k = bit / 32, sh = bit % 32;
      uint64_t v = w[k];
      if (k + 1 < 8) v |= (uint64_t)w[k + 1] << 32;
      a[j] = j < 8 ? (uint32_t)((v >> sh) & 0x1fffffffu) : (uint32_t)(v >> sh);
    }
  }
  // product scanning with interleaved reduction: columns of a * r2 + m * p, 29 bits at a time
  uint32_t m[9], r[9];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 17; k++) {
#pragma unroll
    for (int t = 0; t < 9; t++)
      if (k - t >= 0 && k - t < 9) acc += (uint64_t)a[t] * r2[k - t];
#pragma unroll
    for (int t = 0; t < 9; t++)
      if (t < k && k - t < 9) acc += (uint64_t)m[t] * zkir_p29[k - t];
    if (k < 9) {
      m[k] = ((uint32_t)acc * ZKIR_FR29_NP) & 0x1fffffffu;
      acc += (uint64_t)m[k] * zkir_p29[0];
    } else {
      r[k - 9] = (uint32_t)acc & 0x1fffffffu;
    }
    acc >>= 29;
  }
  r[8] = (uint32_t)acc;
#pragma unroll
  for (int j = 0; j < 9; j++) dst[(size_t)i * 12 + j] = r[j];
  dst[(size_t)i * 12 + 9] = dst[(size_t)i * 12 + 10] = dst[(size_t)i * 12 + 11] = 0;
}

// What a single-tape launch starts from: the caller's four result words {0, all ones, 0, 0} and the work counters
// (64 words) zeroed.  One kernel instead of three memsets: a memset recorded into a HIP graph was seen to write
// other values on the graph's second replay (ROCm 7.2), a kernel node replays with the arguments it was recorded with.
__global__ void zkir_init_launch_kernel(uint64_t* result, uint32_t* counters) {
  if (threadIdx.x < 4 && result) result[threadIdx.x] = threadIdx.x == 1 ? ~0ull : 0ull;
  if (threadIdx.x < 64 && counters) counters[threadIdx.x] = 0;
}

// constants arrive canonical; rewrite them in Montgomery form in place
__global__ void zkir_consts_to_mont_kernel(uint32_t* consts, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr a;
#pragma unroll
  for (int k = 0; k < 8; k++) a.l[k] = consts[(size_t)i * 8 + k];
  a = fr_to_mont(a);
#pragma unroll
  for (int k = 0; k < 8; k++) consts[(size_t)i * 8 + k] = a.l[k];
}

// ---- peak-rate probes ------------------------------------------------------
// Issue rate of one integer multiply instruction: 8 independent dependency
// chains per lane, 16 instructions per loop trip, nothing else in the loop.
template <int WHICH>
__global__ void __launch_bounds__(256) zkir_int_mul_rate_kernel(uint32_t* out, uint32_t seed, int iters) {
  uint32_t a[8], b = seed | 1u, c = seed * 3u + 7u;
  uint64_t q[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    a[i] = threadIdx.x * 8 + i + seed;
    q[i] = a[i];
  }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
#pragma unroll
      for (int n = 0; n < 8; n++) {
        if (WHICH == 0) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[n]) : "v"(b));
        if (WHICH == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[n]) : "v"(b));
        if (WHICH == 2) {
          uint64_t sink;
          asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(q[n]), "=s"(sink) : "v"(b), "v"(c));
        }
      }
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s ^= a[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Montgomery multiplications back to back in registers: two independent chains per lane.
__global__ void __launch_bounds__(256) zkir_fr_mul_rate_kernel(uint32_t* out, uint32_t seed, int iters) {
  Fr x = fr_one_mont(), y = fr_r2();
  x.l[0] ^= threadIdx.x + seed;
  x.l[7] &= 0x0fffffffu;
  y.l[0] ^= blockIdx.x;
  y.l[7] &= 0x0fffffffu;
  for (int it = 0; it < iters; it++) {
    x = fr_mul(x, y);
    y = fr_mul(y, x);
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s ^= x.l[i] ^ y.l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the 29-bit multiplier back to back in registers: 4 * iters products per lane
__global__ void __launch_bounds__(256) zkir_fr29_mul_rate_kernel(uint32_t* out, uint32_t seed_in, int iters) {
  uint32_t seed = seed_in + threadIdx.x * 2654435761u + blockIdx.x, res;
  asm volatile(ZKIR_FR29_RATE_ASM : [res] "=v"(res) : [seed] "v"(seed), [iters] "s"(iters) : ZKIR_INTERP29_CORE_CLOBBERS);
  out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}

}  // namespace zkir

// ---- host-callable launchers (C++ linkage, used by runtime.cpp) ------------
namespace zkir {

// variant: 0 = + - * tapes (compiled), 1 = every operator (compiled), 2 = + - * tapes (generated stream, 8 x 32-bit
// limbs), 3 = + - * tapes (generated stream, 9 x 29-bit limbs)
static const void* interp_entry(int variant) {
  switch (variant) {
    case 1: return (const void*)zkir_interp_kernel<true>;
    case 2: return (const void*)zkir_interp_lean_kernel;
    case 3: return (const void*)zkir_interp_lean29_kernel;
    default: return (const void*)zkir_interp_kernel<false>;
  }
}

// The generated stream parks evicted operands in one extra 8 KiB row of LDS when that row does
// not cost a resident workgroup (five fit a CU up to 32 KiB each).
static bool interp_victim_row(int variant, uint32_t lslots, uint32_t strands) {
  static const bool off = std::getenv("ZKIR_NO_VICTIM_ROW") != nullptr;  // measurement aid
  if (strands || off) return false;
  return (variant == 2 && lslots + 1 <= 4) || (variant == 3 && lslots + 1 <= 3);
}
// dynamic LDS of a workgroup: the window slots, plus the victim row, plus (strand tapes) the
// exchange slots of 64 lanes x 32 bytes
static size_t interp_lds_bytes(int variant, uint32_t lslots, uint32_t strands, uint32_t xslots) {
  const size_t rows = (size_t)lslots * (strands == 8 ? 2 : 1) + (interp_victim_row(variant, lslots, strands) ? 1 : 0);
  if (variant == 3) return kTable29 + rows * kSlot29 + (strands ? (size_t)xslots * kWave29 : 0);
  return rows * 2 * kBlock * sizeof(uint4) + (strands ? (size_t)xslots * 2048 : 0);
}
// what a launch would ask of a CU's 160 KiB of LDS (the host refuses launches beyond it with a message of its own)
size_t interp_lds_need(int variant, uint32_t lslots, uint32_t strands, uint32_t xslots) {
  return interp_lds_bytes(variant, lslots, strands, xslots);
}
// bytes of one witness slot of one workgroup in device memory
size_t interp_slot_bytes(int variant) { return variant == 3 ? kSlot29 : (size_t)2 * kBlock * sizeof(uint4); }

hipError_t launch_interp(int variant, const TapeDesc* tapes_dev, uint32_t n_tapes, uint32_t passes_per_tape,
                         uint64_t n_vectors, const uint32_t* inputs, uint32_t* outputs, void* gslot_pool,
                         uint32_t gslots_per_block, uint32_t lslots, uint64_t* unsat_per_tape, uint64_t* stats,
                         uint32_t* work_counter, uint32_t grid, uint32_t strands, uint32_t xslots, uint32_t n_cu,
                         uint64_t* faults_per_tape, hipStream_t stream) {
  if (n_cu == 0 || n_cu > 0xffffu) return hipErrorInvalidValue;
  const size_t lds_bytes = interp_lds_bytes(variant, lslots, strands, xslots);
  if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
  if ((variant == 2 || variant == 3) && n_vectors * 32 >= (1ull << 32)) return hipErrorInvalidValue;
  if ((variant == 2 || variant == 3) && (uint64_t)gslots_per_block * interp_slot_bytes(variant) >= (1ull << 32)) return hipErrorInvalidValue;
  if (strands && variant != 2 && variant != 3) return hipErrorInvalidValue;  // only the generated streams know strand tapes
  if (strands != 0 && strands != 4 && strands != 8) return hipErrorInvalidValue;
  if (strands == 8) gslots_per_block *= 2;  // two sets of slot rows (a row holds four wavefronts)
  if ((variant == 2 || variant == 3) && (uint64_t)gslots_per_block * interp_slot_bytes(variant) >= (1ull << 32)) return hipErrorInvalidValue;
  if (lds_bytes > 48 * 1024) {  // above the default dynamic-LDS limit the kernel must opt in
    hipError_t e = hipFuncSetAttribute(interp_entry(variant), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  uint4* pool = (uint4*)gslot_pool;
  // the last four are read by the generated-stream kernels only
  uint32_t victim_row = interp_victim_row(variant, lslots, strands) ? 1u : 0u;
  // A strand launch with several workgroups per CU: the workgroups take turns at the SIMDs' issue priority (164 us
  // each) instead of running in the order they were placed, the last one finishing alone.
  static const uint32_t turn_shift = std::getenv("ZKIR_TURN_SHIFT") ? (uint32_t)std::atoi(std::getenv("ZKIR_TURN_SHIFT")) : 14u;  // measurement aid; 99: no turns
  uint32_t turns = strands && grid > n_cu && turn_shift < 32 ? (n_cu | turn_shift << 16) : 0u;
  // the compiled kernels take the per-tape fault counters as their last argument (only they can fault: the
  // generated streams run + - * tapes); the generated-stream kernels take the four that follow
  void* args_compiled[] = {&tapes_dev, &n_tapes, &passes_per_tape, &n_vectors, &inputs, &outputs, &pool, &gslots_per_block,
                           &unsat_per_tape, &stats, &work_counter, &faults_per_tape};
  void* args_stream[] = {&tapes_dev, &n_tapes, &passes_per_tape, &n_vectors, &inputs, &outputs, &pool, &gslots_per_block,
                         &unsat_per_tape, &stats, &work_counter, &victim_row, &strands, &lslots, &turns};
  void** args = (variant == 2 || variant == 3) ? args_stream : args_compiled;
  return hipLaunchKernel(interp_entry(variant), dim3(grid), dim3(strands == 8 ? 2 * kBlock : kBlock), args, lds_bytes, stream);
}

// workgroups of the interpreter that fit on one CU with `lslots` window slots
hipError_t interp_blocks_per_cu(int variant, uint32_t lslots, uint32_t strands, uint32_t xslots, int* blocks) {
  const size_t lds_bytes = interp_lds_bytes(variant, lslots, strands, xslots);
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, interp_entry(variant), strands == 8 ? 2 * kBlock : kBlock, lds_bytes);
  if (e == hipSuccess && *blocks < 1) *blocks = 1;
  return e;
}

hipError_t launch_init(uint64_t* result_dev, uint32_t* counters_dev, hipStream_t stream) {
  hipLaunchKernelGGL(zkir_init_launch_kernel, dim3(1), dim3(64), 0, stream, result_dev, counters_dev);
  return hipGetLastError();
}

hipError_t launch_consts_to_mont(uint32_t* consts_dev, uint32_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(zkir_consts_to_mont_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, consts_dev, n);
  return hipGetLastError();
}

hipError_t launch_consts_to_mont29(const uint32_t* canonical_dev, uint32_t* dst_dev, uint32_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(zkir_consts_to_mont29_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, canonical_dev, dst_dev, n);
  return hipGetLastError();
}

hipError_t launch_int_mul_rate(int which, uint32_t* out, uint32_t blocks, int iters, hipStream_t stream) {
  switch (which) {
    case 0: hipLaunchKernelGGL(zkir_int_mul_rate_kernel<0>, dim3(blocks), dim3(256), 0, stream, out, 12345u, iters); break;
    case 1: hipLaunchKernelGGL(zkir_int_mul_rate_kernel<1>, dim3(blocks), dim3(256), 0, stream, out, 12345u, iters); break;
    default: hipLaunchKernelGGL(zkir_int_mul_rate_kernel<2>, dim3(blocks), dim3(256), 0, stream, out, 12345u, iters); break;
  }
  return hipGetLastError();
}

hipError_t launch_fr_mul_rate(uint32_t* out, uint32_t blocks, int iters, hipStream_t stream) {
  hipLaunchKernelGGL(zkir_fr_mul_rate_kernel, dim3(blocks), dim3(256), 0, stream, out, 777u, iters);
  return hipGetLastError();
}

// 2 * iters products per lane like launch_fr_mul_rate (iters even)
hipError_t launch_fr29_mul_rate(uint32_t* out, uint32_t blocks, int iters, hipStream_t stream) {
  hipLaunchKernelGGL(zkir_fr29_mul_rate_kernel, dim3(blocks), dim3(256), 0, stream, out, 777u, iters / 2);
  return hipGetLastError();
}

}  // namespace zkir
