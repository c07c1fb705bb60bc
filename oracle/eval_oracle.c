/*
 * eval_oracle.c - scalar CPU interpreter of zkIR tapes (plain C, 4 x 64-bit limbs).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker the GPU kernels are compared
 * with and the CPU baseline bench.py times; it is never linked into, loaded by or
 * called from the shipped library (zkir_amd/libzkir_hip.so).
 *
 * PARITY UNPINNED against circom: the reference (razn-v/zkir) has no evaluator
 * (`Field` is a `String`, src/ast.rs:12) and no test vectors; the operator meanings
 * below restate circom's (the tokens src/circom.rs:35-157 prints).  What pins this
 * file is tests/golden/fr_kat.json (Python big-integer known answers) and the
 * cross-check against oracle/fr_ref.py.
 *
 * Written independently of the device code: 64-bit limbs, unsigned __int128
 * products, word-serial Montgomery reduction; the kernels use 32-bit limbs and a
 * fixed instruction stream.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t w[4]; } fe;

/* tape format: see include/zkir_hip.h (restated here so the oracle has no dependency on the product tree) */
typedef struct { uint32_t code, dst, a, b; } op_t;
typedef struct {
  uint32_t magic, version, n_inputs, n_outputs, n_consts, n_ops, n_gslots, n_lslots, n_constraints, n_mul_ops, flags,
      strand_ops[8], n_xslots, n_strands, n_scale_ops;
} hdr_t;
enum { K_GSLOT = 0, K_LSLOT = 1, K_CONST = 2, K_INPUT = 3, K_ACC = 4, K_NONE = 5, K_XSLOT = 6 };
enum { F_STRANDS = 1, STRANDS = 8 };  /* STRANDS: the most a tape can have (n_strands is 4 or 8) */
enum {
  OP_NOP = 0, OP_INPUT = 1, OP_OUTPUT = 2, OP_COPY = 3, OP_NEG = 4, OP_BNOT = 5, OP_BNOT256 = 6, OP_BARRIER = 7,
  OP_INPUT_SCALED = 8, OP_OUTPUT_SCALED = 9, /* the input times 2^-256, the output times 2^256 (include/zkir_hip.h) */
  OP_ADD = 16, OP_SUB, OP_MUL, OP_POW, OP_DIV, OP_IDIV, OP_REM, OP_LT, OP_LE, OP_GT, OP_GE, OP_EQ, OP_AND, OP_OR,
  OP_BAND, OP_BOR, OP_BXOR, OP_SHR, OP_SHL,
  OP_CHECK = 48
};

static const fe P = {{0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull}};
static const fe HALF = {{0xa1f0fac9f8000000ull, 0x9419f4243cdcb848ull, 0xdc2822db40c0ac2eull, 0x183227397098d014ull}};
static const fe R2 = {{0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull}};
static const fe ONE_M = {{0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full}};
static const uint64_t NP = 0xc2e1f593efffffffull; /* -p^-1 mod 2^64 */

static int is_zero(const fe* a) { return (a->w[0] | a->w[1] | a->w[2] | a->w[3]) == 0; }
static int eq(const fe* a, const fe* b) {
  return ((a->w[0] ^ b->w[0]) | (a->w[1] ^ b->w[1]) | (a->w[2] ^ b->w[2]) | (a->w[3] ^ b->w[3])) == 0;
}
static uint64_t add_raw(fe* r, const fe* a, const fe* b) {
  u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)a->w[i] + b->w[i]; r->w[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
static uint64_t sub_raw(fe* r, const fe* a, const fe* b) {
  uint64_t bw = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a->w[i] - b->w[i] - bw; r->w[i] = (uint64_t)d; bw = (uint64_t)(d >> 64) & 1; }
  return bw;
}
static int lt_raw(const fe* a, const fe* b) { fe d; return (int)sub_raw(&d, a, b); }
static void reduce_once(fe* x) { fe d; if (!sub_raw(&d, x, &P)) *x = d; }

static void f_add(fe* r, const fe* a, const fe* b) { add_raw(r, a, b); reduce_once(r); }
static void f_sub(fe* r, const fe* a, const fe* b) { fe d; if (sub_raw(&d, a, b)) add_raw(&d, &d, &P); *r = d; }
static void f_neg(fe* r, const fe* a) { if (is_zero(a)) *r = *a; else sub_raw(r, &P, a); }

/* Montgomery product a*b/2^256 mod p: full 512-bit product, then four word reductions */
static void f_mul(fe* r, const fe* a, const fe* b) {
  uint64_t t[9] = {0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a->w[i] * b->w[j] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
    t[i + 4] = (uint64_t)c;
  }
  for (int i = 0; i < 4; i++) {
    const uint64_t m = t[i] * NP;
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)m * P.w[j] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
    for (int k = i + 4; k < 9 && c; k++) { c += t[k]; t[k] = (uint64_t)c; c >>= 64; }
  }
  fe x = {{t[4], t[5], t[6], t[7]}};
  reduce_once(&x); /* the value is below 2p */
  *r = x;
}
static void to_mont(fe* r, const fe* a) { f_mul(r, a, &R2); }
static void from_mont(fe* r, const fe* a) { fe one = {{1, 0, 0, 0}}; f_mul(r, a, &one); }

static void f_pow(fe* r, const fe* a_m, const fe* e) { /* a in Montgomery form, e canonical */
  fe acc = ONE_M;
  for (int bit = 253; bit >= 0; bit--) {
    f_mul(&acc, &acc, &acc);
    if ((e->w[bit >> 6] >> (bit & 63)) & 1) f_mul(&acc, &acc, a_m);
  }
  *r = acc;
}

/* canonical integer division by shift and subtract; b != 0 */
static void divrem(const fe* a, const fe* b, fe* q, fe* rem) {
  fe r = {{0, 0, 0, 0}}, qq = {{0, 0, 0, 0}};
  for (int bit = 255; bit >= 0; bit--) {
    for (int i = 3; i > 0; i--) r.w[i] = (r.w[i] << 1) | (r.w[i - 1] >> 63);
    r.w[0] = (r.w[0] << 1) | ((a->w[bit >> 6] >> (bit & 63)) & 1);
    fe d;
    if (!sub_raw(&d, &r, b)) { r = d; qq.w[bit >> 6] |= 1ull << (bit & 63); }
  }
  *q = qq;
  *rem = r;
}
static int signed_lt(const fe* a, const fe* b) {
  const int na = lt_raw(&HALF, a), nb = lt_raw(&HALF, b);
  if (na != nb) return na;
  return lt_raw(a, b);
}
static void shr_bits(fe* r, const fe* a, unsigned k) { /* k < 256 */
  fe o = {{0, 0, 0, 0}};
  const unsigned ws = k >> 6, bs = k & 63;
  for (unsigned i = 0; i + ws < 4; i++) {
    uint64_t v = a->w[i + ws] >> bs;
    if (bs && i + ws + 1 < 4) v |= a->w[i + ws + 1] << (64 - bs);
    o.w[i] = v;
  }
  *r = o;
}
static void shl_bits_254(fe* r, const fe* a, unsigned k) { /* ((a << k) & (2^254-1)) mod p */
  fe o = {{0, 0, 0, 0}};
  const unsigned ws = k >> 6, bs = k & 63;
  for (unsigned i = ws; i < 4; i++) {
    uint64_t v = a->w[i - ws] << bs;
    if (bs && i - ws >= 1) v |= a->w[i - ws - 1] >> (64 - bs);
    o.w[i] = v;
  }
  o.w[3] &= 0x3fffffffffffffffull;
  reduce_once(&o);
  *r = o;
}
static int small_amount(const fe* k) { return (k->w[1] | k->w[2] | k->w[3]) || k->w[0] >= 254 ? -1 : (int)k->w[0]; }
static void f_shift(fe* r, const fe* a, const fe* k, int right) {
  fe amount = *k, zero = {{0, 0, 0, 0}};
  if (lt_raw(&HALF, k)) { sub_raw(&amount, &P, k); right = !right; }
  const int s = small_amount(&amount);
  if (s < 0) { *r = zero; return; }
  if (right) shr_bits(r, a, (unsigned)s); else shl_bits_254(r, a, (unsigned)s);
}

/* binary / unary operators on Montgomery-form values; returns 1 on a division by zero */
static int apply(uint32_t code, fe* r, const fe* x, const fe* y) {
  fe a, b, t, zero = {{0, 0, 0, 0}};
  int flag = -1;
  switch (code) {
    case OP_COPY: *r = *x; return 0;
    case OP_NEG: f_neg(r, x); return 0;
    case OP_ADD: f_add(r, x, y); return 0;
    case OP_SUB: f_sub(r, x, y); return 0;
    case OP_MUL: f_mul(r, x, y); return 0;
    case OP_POW: from_mont(&b, y); f_pow(r, x, &b); return 0;
    case OP_DIV: {
      if (is_zero(y)) { *r = zero; return 1; }
      fe e = P; e.w[0] -= 2; /* Fermat inverse */
      f_pow(&t, y, &e); f_mul(r, x, &t); return 0;
    }
    default: break;
  }
  from_mont(&a, x);
  if (y) from_mont(&b, y); else b = zero;
  switch (code) {
    case OP_LT: flag = signed_lt(&a, &b); break;
    case OP_LE: flag = !signed_lt(&b, &a); break;
    case OP_GT: flag = signed_lt(&b, &a); break;
    case OP_GE: flag = !signed_lt(&a, &b); break;
    case OP_EQ: flag = eq(&a, &b); break;
    case OP_AND: flag = !is_zero(&a) && !is_zero(&b); break;
    case OP_OR: flag = !is_zero(&a) || !is_zero(&b); break;
    default: break;
  }
  if (flag >= 0) { *r = flag ? ONE_M : zero; return 0; }
  switch (code) {
    case OP_IDIV: case OP_REM: {
      if (is_zero(&b)) { *r = zero; return 1; }
      fe q, m; divrem(&a, &b, &q, &m); t = code == OP_IDIV ? q : m; break;
    }
    case OP_BAND: for (int i = 0; i < 4; i++) t.w[i] = a.w[i] & b.w[i]; break;
    case OP_BOR: for (int i = 0; i < 4; i++) t.w[i] = a.w[i] | b.w[i]; reduce_once(&t); break;
    case OP_BXOR: for (int i = 0; i < 4; i++) t.w[i] = a.w[i] ^ b.w[i]; reduce_once(&t); break;
    case OP_BNOT: for (int i = 0; i < 4; i++) t.w[i] = ~a.w[i]; t.w[3] &= 0x3fffffffffffffffull; reduce_once(&t); break;
    case OP_BNOT256: { for (int i = 0; i < 4; i++) t.w[i] = ~a.w[i]; fe q, m; divrem(&t, &P, &q, &m); t = m; break; }
    case OP_SHR: f_shift(&t, &a, &b, 1); break;
    case OP_SHL: f_shift(&t, &a, &b, 0); break;
    default: t = zero; break;
  }
  to_mont(r, &t);
  return 0;
}

typedef struct {
  const hdr_t* h; const op_t* ops; const fe* consts_m;
  const uint32_t* in; uint32_t* out; uint64_t N, lo, hi;
  uint64_t unsat, faults, first_vec, first_con;
} job_t;

static void load_soa(fe* r, const uint32_t* base, uint64_t N, uint64_t elem, uint64_t t) {
  const uint32_t* q = base + elem * 8 * N + t;
  for (int i = 0; i < 4; i++) r->w[i] = (uint64_t)q[(2 * i) * N] | ((uint64_t)q[(2 * i + 1) * N] << 32);
}
static void store_soa(uint32_t* base, uint64_t N, uint64_t elem, uint64_t t, const fe* v) {
  uint32_t* q = base + elem * 8 * N + t;
  for (int i = 0; i < 4; i++) { q[(2 * i) * N] = (uint32_t)v->w[i]; q[(2 * i + 1) * N] = (uint32_t)(v->w[i] >> 32); }
}

/*
 * One vector at a time.  A plain tape is one op stream.  A strand tape (flags & F_STRANDS)
 * is STRANDS op streams with private accumulators and private G / L slots that share the
 * exchange slots x[]; they are run here one after the other from barrier to barrier, which
 * is one of the interleavings the tape's rules allow (a strand reads an exchange slot only
 * after the barrier that follows its write), so any correct parallel run gives the same values.
 * The constraint number of a CHECK is (code >> 8) - 1 when the op carries one (every strand tape, reordered plain
 * tapes), else its position among the CHECK ops.
 */
static void* run_range(void* arg) {
  job_t* j = (job_t*)arg;
  const hdr_t* h = j->h;
  const int strands = (h->flags & F_STRANDS) ? (int)h->n_strands : 1;
  fe* gbuf = (fe*)malloc(sizeof(fe) * ((size_t)h->n_gslots + 1) * (size_t)strands);
  fe lbuf[STRANDS][16], accs[STRANDS], x[64], one = {{1, 0, 0, 0}};
  uint32_t begin[STRANDS + 1];
  begin[0] = 0;
  for (int s = 0; s < strands; s++) begin[s + 1] = strands == 1 ? h->n_ops : begin[s] + h->strand_ops[s];
  j->unsat = j->faults = 0;
  j->first_vec = j->first_con = ~0ull;
  for (uint64_t t = j->lo; t < j->hi; t++) {
    uint64_t k = 0, first_k = ~0ull;
    int faulted = 0, running = strands;
    uint32_t pcs[STRANDS];
    for (int s = 0; s < strands; s++) { pcs[s] = begin[s]; memset(&accs[s], 0, sizeof(fe)); }
    while (running) {
      running = 0;
      for (int s = 0; s < strands; s++) {
        fe* g = gbuf + (size_t)s * ((size_t)h->n_gslots + 1);
        fe* l = lbuf[s];
        fe acc = accs[s];
        uint32_t pc = pcs[s];
        for (; pc < begin[s + 1]; pc++) {
          const op_t* op = &j->ops[pc];
          const uint32_t code = op->code & 0xff;
#define SRC(o) ((o) >> 29 == K_GSLOT ? &g[(o) & 0x1fffffff] : (o) >> 29 == K_LSLOT ? &l[(o) & 0x1fffffff] : \
                (o) >> 29 == K_CONST ? &j->consts_m[(o) & 0x1fffffff] : (o) >> 29 == K_XSLOT ? &x[(o) & 63] : &acc)
          if (code == OP_NOP) continue;
          if (code == OP_BARRIER) { pc++; break; }
          if (code == OP_INPUT) {
            fe v; load_soa(&v, j->in, j->N, op->a & 0x1fffffff, t); to_mont(&acc, &v);
          } else if (code == OP_INPUT_SCALED) {
            /* values are held here as x * 2^256 mod p (Montgomery form with the same radix): the tape value
               input * 2^-256 is held as the input's own words */
            load_soa(&acc, j->in, j->N, op->a & 0x1fffffff, t);
          } else if (code == OP_OUTPUT) {
            if (j->out) { fe v; f_mul(&v, SRC(op->a), &one); store_soa(j->out, j->N, op->dst, t, &v); }
            continue;
          } else if (code == OP_OUTPUT_SCALED) {
            if (j->out) store_soa(j->out, j->N, op->dst, t, SRC(op->a));  /* (a * 2^256) is what is held */
            continue;
          } else if (code == OP_CHECK) {
            fe v; f_mul(&v, SRC(op->a), SRC(op->b));
            const uint64_t id = (op->code >> 8) ? (uint64_t)(op->code >> 8) - 1 : k;
            if (!eq(&v, SRC(op->dst))) {
              j->unsat++;
              if (id < first_k) first_k = id;
            }
            k++;
            continue;
          } else {
            fe r;
            faulted |= apply(code, &r, SRC(op->a), code >= OP_ADD ? SRC(op->b) : NULL);
            acc = r;
          }
          if (op->dst >> 29 == K_GSLOT) g[op->dst & 0x1fffffff] = acc;
          else if (op->dst >> 29 == K_LSLOT) l[op->dst & 0x1fffffff] = acc;
          else if (op->dst >> 29 == K_XSLOT) x[op->dst & 63] = acc;
        }
        accs[s] = acc;
        pcs[s] = pc;
        if (pc < begin[s + 1]) running = 1;
      }
    }
    if (first_k != ~0ull && j->first_vec == ~0ull) { j->first_vec = t; j->first_con = first_k; }
    j->faults += (uint64_t)faulted;
  }
  free(gbuf);
  return NULL;
}

/*
 * Evaluate one tape for N vectors on n_threads host threads (vectors are split into
 * contiguous ranges).  Same buffer layout and result meaning as zkir_eval_batch of
 * include/zkir_hip.h.  The tape is trusted to be well-formed (tests build it with the
 * product's lower, which validates it).  Returns 0, or -1 on a bad argument.
 */
int zkir_eval_cpu(const void* hdr, const void* blob, const uint32_t* inputs_soa, uint64_t N, uint32_t* outputs_soa,
                  uint64_t* unsat_count, uint32_t* first_unsat, uint64_t* faults, int n_threads) {
  const hdr_t* h = (const hdr_t*)hdr;
  if (!h || h->n_lslots > 16 || h->n_xslots > 64 || n_threads < 1 || n_threads > 1024) return -1;
  if ((h->flags & F_STRANDS) && h->n_strands != 4 && h->n_strands != 8) return -1;
  const op_t* ops = (const op_t*)blob;
  const uint32_t* craw = (const uint32_t*)((const char*)blob + (size_t)h->n_ops * 16);
  fe* cm = (fe*)malloc(sizeof(fe) * (h->n_consts + 1));
  for (uint32_t i = 0; i < h->n_consts; i++) {
    fe c;
    for (int k = 0; k < 4; k++) c.w[k] = (uint64_t)craw[i * 8 + 2 * k] | ((uint64_t)craw[i * 8 + 2 * k + 1] << 32);
    to_mont(&cm[i], &c);
  }
  job_t* jobs = (job_t*)calloc((size_t)n_threads, sizeof(job_t));
  pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
  const uint64_t per = (N + (uint64_t)n_threads - 1) / (uint64_t)n_threads;
  for (int i = 0; i < n_threads; i++) {
    job_t* j = &jobs[i];
    j->h = h; j->ops = ops; j->consts_m = cm; j->in = inputs_soa; j->out = outputs_soa; j->N = N;
    j->lo = per * (uint64_t)i < N ? per * (uint64_t)i : N;
    j->hi = j->lo + per < N ? j->lo + per : N;
    if (n_threads == 1) run_range(j); else pthread_create(&th[i], NULL, run_range, j);
  }
  uint64_t u = 0, f = 0, fv = ~0ull, fc = ~0ull;
  for (int i = 0; i < n_threads; i++) {
    if (n_threads > 1) pthread_join(th[i], NULL);
    u += jobs[i].unsat; f += jobs[i].faults;
    if (jobs[i].first_vec < fv) { fv = jobs[i].first_vec; fc = jobs[i].first_con; }
  }
  if (unsat_count) *unsat_count = u;
  if (faults) *faults = f;
  if (first_unsat) { first_unsat[0] = fv == ~0ull ? 0xffffffffu : (uint32_t)fv; first_unsat[1] = fv == ~0ull ? 0xffffffffu : (uint32_t)fc; }
  free(jobs); free(th); free(cm);
  return 0;
}

/* single operator on canonical operands (for the known-answer tests): r = a <op> b */
int zkir_oracle_op(uint32_t code, const uint32_t a8[8], const uint32_t b8[8], uint32_t r8[8]) {
  fe a, b, am, bm, rm, r;
  for (int k = 0; k < 4; k++) {
    a.w[k] = (uint64_t)a8[2 * k] | ((uint64_t)a8[2 * k + 1] << 32);
    b.w[k] = (uint64_t)b8[2 * k] | ((uint64_t)b8[2 * k + 1] << 32);
  }
  to_mont(&am, &a); to_mont(&bm, &b);
  const int fault = apply(code, &rm, &am, code >= OP_ADD ? &bm : NULL);
  from_mont(&r, &rm);
  for (int k = 0; k < 4; k++) { r8[2 * k] = (uint32_t)r.w[k]; r8[2 * k + 1] = (uint32_t)(r.w[k] >> 32); }
  return fault;
}
