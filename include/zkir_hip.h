/*
 * zkir_hip.h - C ABI of the MI355X (gfx950) batched evaluator for zkIR circuits.
 *
 * The reference (razn-v/zkir) is a Rust crate whose only backend is the Circom
 * pretty-printer: a struct that borrows `&Circuit` and exposes one method
 * (src/circom.rs:4-15, :412).  This library is a second backend behind the
 * same shape: take a Circuit (src/ast.rs:113-116), lower it to a flat tape,
 * and evaluate the tape for many input vectors on the GPU.  Every entry point
 * names the reference item it mirrors or, where the reference has nothing
 * (it has no evaluator: `Field` is a `String`, src/ast.rs:12), says so.
 *
 * Conventions
 *   - every function returns int32_t: 0 = ZKIR_OK, negative = error code;
 *     nothing throws or aborts across this boundary;
 *   - the caller owns every buffer it passes; handles are opaque;
 *   - one zkir_ctx per device; a context is not thread-safe, separate
 *     contexts are independent.  The evaluation calls on device pointers are
 *     asynchronous, and the launches of ONE context share its work counters,
 *     descriptor buffer and witness-slot pool: the library makes each launch
 *     wait (hipStreamWaitEvent) for the previous launch of the same context,
 *     so calls on different streams give the results of serial calls, they
 *     just do not overlap.  Use one context per stream for overlap;
 *   - field elements crossing the ABI are canonical integers in [0, p) of the
 *     BN254 scalar field, 8 little-endian 32-bit limbs; Montgomery form is
 *     internal to the device code;
 *   - batches are limb-major structure-of-arrays: element `sig` of vector `t`
 *     has limb `l` at  buf[(sig*8 + l)*N + t]  (lane t reads consecutive
 *     addresses: coalesced).
 */
#ifndef ZKIR_HIP_H
#define ZKIR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ errors */
#define ZKIR_OK 0
#define ZKIR_E_INVALID (-1)   /* bad argument / malformed tape or text        */
#define ZKIR_E_NOMEM (-2)     /* host or device allocation failed            */
#define ZKIR_E_HIP (-3)       /* a HIP runtime call failed                   */
#define ZKIR_E_NODEVICE (-4)  /* no usable gfx950 device                     */
#define ZKIR_E_LOWER (-5)     /* circuit rejected by lower; see status class */
#define ZKIR_E_BUFSIZE (-6)   /* caller buffer too small; *needed is set     */
#define ZKIR_E_LIMIT (-7)     /* size beyond a documented maximum            */
#define ZKIR_E_INTERNAL (-8)  /* a host-side exception was caught at the ABI */
#define ZKIR_E_UNSUPPORTED (-9) /* the circuit uses a construct this backend cannot express */
#define ZKIR_E_BUSY (-10)     /* a shared resource is in use by another context's launch */

/* ------------------------------------------------------------- tape format */
/*
 * A tape is the flat, loop-free form of one Circuit: the reference has no such
 * thing; it is what its Instruction tree (src/ast.rs:91-111) becomes once the
 * For loops (src/generator.rs:401-456) are unrolled and every Expr node
 * (src/ast.rs:15-42) is one op.
 *
 * blob layout (all little-endian, 4-byte aligned):
 *     zkir_op   ops[n_ops]
 *     uint32_t  consts[n_consts][8]        canonical limbs
 *     uint32_t  triples[n_constraints][3]  operands (a, b, c): w[a]*w[b]==w[c]
 *
 * Strand tapes (flags & ZKIR_TAPE_F_STRANDS).  A circuit whose ops are not one
 * dependent chain can be cut into n_strands (4 or 8) op streams that run side by side, one
 * wavefront each, on the SAME 64 input vectors (instead of four wavefronts on 256
 * vectors): the shape for batches too small to fill the machine; the smaller the batch,
 * the more strands pay.  ops[] then holds
 * strand 0's ops, then strand 1's, ... (strand_ops[] each).  Every strand has its own
 * accumulator and its own GSLOT / LSLOT slots; values cross between strands only through
 * XSLOT operands, and only across a ZKIR_OP_BARRIER: a strand may read an exchange slot
 * another strand wrote only after both have passed the same barrier, and may overwrite
 * it only after a barrier that every reader of the old value has reached.  All strands
 * hold the same number of barriers.  Constraint numbers ride in the CHECK ops
 * (bits 8..31 of `code` = constraint number + 1), since a strand sees only its own checks.
 */
#define ZKIR_TAPE_F_STRANDS 1u
#define ZKIR_STRANDS 4u     /* strands of a strand tape: this many ... */
#define ZKIR_MAX_STRANDS 8u /* ... or this many (zkir_tape_hdr.n_strands) */
#define ZKIR_MAX_XSLOTS 48u
#define ZKIR_TAPE_MAGIC 0x31544b5au /* "ZKT1" */
#define ZKIR_TAPE_VERSION 2u

typedef struct zkir_op {
  uint32_t code; /* zkir_opcode in bits 0..7; bits 8..31: 0, or in a strand
                    tape's CHECK op the constraint number + 1                */
  uint32_t dst;  /* operand (kind GSLOT / LSLOT / NONE), or a raw index for
                    ZKIR_OP_OUTPUT (output element) and the `c` operand for
                    ZKIR_OP_CHECK                                            */
  uint32_t a;    /* operand                                                  */
  uint32_t b;    /* operand, ZKIR_OPERAND_NONE for unary ops                 */
} zkir_op;

/* operand = kind << 29 | index */
#define ZKIR_KIND_SHIFT 29
#define ZKIR_INDEX_MASK 0x1fffffffu
#define ZKIR_KIND_GSLOT 0u /* witness slot kept in device memory (coalesced)  */
#define ZKIR_KIND_LSLOT 1u /* witness slot kept in the on-chip window (LDS)   */
#define ZKIR_KIND_CONST 2u /* constant-pool entry                             */
#define ZKIR_KIND_INPUT 3u /* input element (only as `a` of ZKIR_OP_INPUT)    */
#define ZKIR_KIND_ACC 4u   /* the previous op's result, still in registers    */
#define ZKIR_KIND_NONE 5u  /* no operand / result is not stored               */
#define ZKIR_KIND_XSLOT 6u /* exchange slot shared by the strands (strand tapes) */
#define ZKIR_OPERAND(kind, idx) (((uint32_t)(kind) << ZKIR_KIND_SHIFT) | ((uint32_t)(idx) & ZKIR_INDEX_MASK))
#define ZKIR_OPERAND_NONE ZKIR_OPERAND(ZKIR_KIND_NONE, 0)
#define ZKIR_OPERAND_ACC ZKIR_OPERAND(ZKIR_KIND_ACC, 0)

/*
 * Opcodes.  The arithmetic meaning of each is the Circom operator that
 * src/circom.rs prints for the corresponding Expr variant (line cited).
 */
typedef enum zkir_opcode {
  ZKIR_OP_NOP = 0,
  /* unary / data movement */
  ZKIR_OP_INPUT = 1,   /* dst = inputs[a]                (signal input, circom.rs:219) */
  ZKIR_OP_OUTPUT = 2,  /* outputs[dst] = a               (signal output, circom.rs:220) */
  ZKIR_OP_COPY = 3,    /* dst = a                        (Assign / `<==` of a leaf)     */
  ZKIR_OP_NEG = 4,     /* dst = -a mod p                 (emitted by lower only)        */
  ZKIR_OP_BNOT = 5,    /* ~a over 254 bits, then mod p   (circom.rs:136)                */
  ZKIR_OP_BNOT256 = 6, /* ~a over 256 bits, then mod p   (alternative reading of `~`)   */
  ZKIR_OP_BARRIER = 7, /* strand tapes: wait for the other strands (no operands)        */
  /* Scaled values.  lower may keep a value v of the circuit as the tape value v * 2^-256 mod p ("scaled"): the
     product of a scaled and a plain value is scaled, sums of scaled values are scaled, and the two ops below take
     scaled values in and out of a tape at no cost on the device, which holds a tape value x as the words of
     x * 2^256 mod p: the words of a scaled value are those of the circuit's value itself.  A circuit whose every
     signal is an output then pays no conversion per output, only a product by the constant 2^256 for the signals
     that are needed as the plain factor of a product.  To an interpreter of tapes these are just two more ops.  */
  ZKIR_OP_INPUT_SCALED = 8,  /* dst = inputs[a] * 2^-256 mod p                          */
  ZKIR_OP_OUTPUT_SCALED = 9, /* outputs[dst] = a * 2^256 mod p                          */
  /* binary */
  ZKIR_OP_ADD = 16,  /* circom.rs:35-41   */
  ZKIR_OP_SUB = 17,  /* circom.rs:42-46   */
  ZKIR_OP_MUL = 18,  /* circom.rs:47-51   */
  ZKIR_OP_POW = 19,  /* circom.rs:52-56   */
  ZKIR_OP_DIV = 20,  /* circom.rs:57-61   field division; b == 0 gives 0 and a fault */
  ZKIR_OP_IDIV = 21, /* circom.rs:62-66   integer quotient; b == 0: 0 and a fault    */
  ZKIR_OP_REM = 22,  /* circom.rs:67-71   integer remainder; b == 0: 0 and a fault   */
  ZKIR_OP_LT = 23,   /* circom.rs:72-78   signed representatives                     */
  ZKIR_OP_LE = 24,   /* circom.rs:79-85   */
  ZKIR_OP_GT = 25,   /* circom.rs:86-92   */
  ZKIR_OP_GE = 26,   /* circom.rs:93-99   */
  ZKIR_OP_EQ = 27,   /* circom.rs:100-106 */
  ZKIR_OP_AND = 28,  /* circom.rs:107-113 logical */
  ZKIR_OP_OR = 29,   /* circom.rs:114-120 logical */
  ZKIR_OP_BAND = 30, /* circom.rs:122-128 */
  ZKIR_OP_BOR = 31,  /* circom.rs:129-135 */
  ZKIR_OP_BXOR = 32, /* circom.rs:137-143 */
  ZKIR_OP_SHR = 33,  /* circom.rs:144-150 */
  ZKIR_OP_SHL = 34,  /* circom.rs:151-157 */
  /* constraint: count (vector, constraint) pairs with w[a]*w[b] != w[dst]
     (`<==`, circom.rs:201-214); the k-th CHECK op of a tape is constraint k.
     When b is the tape's constant 1 (lower writes the constraint of a signal
     whose value is linear as `value * 1 == signal`) the device compares w[a]
     with w[dst] without multiplying: x * 1 is x, so no result differs.     */
  ZKIR_OP_CHECK = 48,
  ZKIR_OP__MAX = 49
} zkir_opcode;

typedef struct zkir_tape_hdr {
  uint32_t magic;
  uint32_t version;
  uint32_t n_inputs;      /* input field elements per vector (arrays flattened) */
  uint32_t n_outputs;     /* output field elements per vector                   */
  uint32_t n_consts;
  uint32_t n_ops;
  uint32_t n_gslots;      /* device-memory witness slots                        */
  uint32_t n_lslots;      /* on-chip window slots (<= ZKIR_MAX_LSLOTS)          */
  uint32_t n_constraints; /* number of ZKIR_OP_CHECK ops                        */
  uint32_t n_mul_ops;     /* ZKIR_OP_MUL ops (work accounting only)             */
  uint32_t flags;         /* ZKIR_TAPE_F_*                                      */
  uint32_t strand_ops[8]; /* strand tapes: ops of each strand (sum = n_ops); else 0 */
  uint32_t n_xslots;      /* strand tapes: exchange slots (<= ZKIR_MAX_XSLOTS)  */
  uint32_t n_strands;     /* strand tapes: ZKIR_STRANDS or ZKIR_MAX_STRANDS; else 0 */
  uint32_t n_scale_ops;   /* of the ZKIR_OP_MUL ops, those lower added to move a value between its scaled and its
                             plain form (products by the constants 2^256 or 2^-256): not the circuit's own
                             multiplications (work accounting only, like n_mul_ops)  */
} zkir_tape_hdr; /* 88 bytes; n_gslots / n_lslots of a strand tape are per strand */

#define ZKIR_MAX_LSLOTS 16u

/* -------------------------------------------------- reference-side mirror */
/*
 * Host-side objects that mirror the reference's own types, so a campaign
 * written against the Rust crate maps one to one:
 *   zkir_circuit   <-> ast::Circuit          (src/ast.rs:113-132)
 *   zkir_generate  <-> Rng + Generator::new + Generator::generate
 *                                            (src/rng.rs:11-53, src/generator.rs:124-164)
 *   zkir_circuit_transpile_circom <-> CircomTranspiler::new + transpile_circuit
 *                                            (src/circom.rs:10-15, :412-429)
 * Circuits cross the ABI as an s-expression text form of the AST (grammar at
 * the top of zkir_amd/csrc/host/sexpr.hpp); the Rust side would serialise its
 * own `Circuit` to it.
 */
typedef struct zkir_circuit zkir_circuit;
typedef struct zkir_tape zkir_tape;

/* Knobs of the generator.  zkir_generator_config_default() fills in the
 * literals hard-coded in the reference, so the default configuration draws
 * exactly the reference's random stream. */
typedef struct zkir_generator_config {
  uint32_t min_instructions;   /* 15  src/generator.rs:147 */
  uint32_t max_instructions;   /* 30  src/generator.rs:147 */
  uint32_t statement_budget;   /* 10  src/generator.rs:167 */
  uint32_t max_array_len;      /* 10  src/generator.rs:224 */
  uint32_t max_expr_depth;     /* 1   src/generator.rs:486 */
  uint32_t loop_init_max;      /* 30  src/generator.rs:406 */
  uint32_t loop_bound_max;     /* 60  src/generator.rs:416 */
  uint32_t loop_body_max;      /* 5   src/generator.rs:436 */
  uint32_t name_len;           /* 8   src/generator.rs:264 */
  /* additive extensions (all 0 in the default configuration) */
  uint32_t inputs_readable;    /* 1: input signals may be read by expressions */
  uint32_t arithmetic_only;    /* 1: expressions use only + - *               */
  uint32_t quadratic_aware;    /* 1: `<==` right-hand sides are lin*lin+lin   */
  uint32_t forced_inputs;      /* declare this many scalar inputs up front    */
  uint32_t no_loops;           /* 1: never emit For                           */
  uint32_t safe_array_index;   /* 1: array reads name a written element (the
                                  reference draws a position in 0..=len,
                                  src/generator.rs:584, which can be unwritten
                                  or one past the end)                        */
  uint32_t signal_window;      /* quadratic_aware: linear terms draw among the inputs
                                  and the last `signal_window` readable signals
                                  instead of all of them (0 = all)            */
  uint32_t enable_if;          /* 0: never emit If, as the reference ships
                                  (src/generator.rs:190 is commented out);
                                  1: dispatch gen_if (src/generator.rs:355-399)
                                  as written, conditions are random expressions;
                                  2: the same with conditions that compare two
                                  constants, so they are known when lowering  */
  uint32_t max_linear_terms;   /* quadratic_aware: each of the three linear parts of
                                  `lin*lin+lin` has up to this many terms (1 or 2;
                                  0 means 2)                                    */
  uint32_t signal_factors;     /* quadratic_aware, 1: every term of the two factors of
                                  `lin*lin+lin` is a signal (the draw that would make
                                  it a constant is not made), so every constraint has
                                  two signal factors; the addend draws as before   */
  uint32_t reserved[1];
} zkir_generator_config;

int32_t zkir_generator_config_default(zkir_generator_config* cfg);

/* Rng::new + Generator::new + generate (src/main.rs:81-83) with an explicit
 * seed instead of the timestamp counter (src/rng.rs:14-16).  seed 0 is a fixed
 * point of the xorshift (src/rng.rs:18-24) and is rejected.  cfg may be NULL. */
int32_t zkir_generate(uint64_t seed, const zkir_generator_config* cfg, zkir_circuit** out);

/* The two counts generate() prints while it runs, "Generating {} instructions..." and
 * "Generated {}/{} instructions" (src/generator.rs:148, :158): instructions asked for
 * and top-level instructions produced.  Zero for circuits that were parsed, not generated. */
int32_t zkir_circuit_generation_counts(const zkir_circuit* c, uint64_t* requested, uint64_t* generated);

/* The first `n` values Rng::next would return for `seed` (src/rng.rs:18-24). */
int32_t zkir_rng_stream(uint64_t seed, uint32_t n, uint64_t* out);

/* Build a Circuit from / print it to the s-expression text form. */
int32_t zkir_circuit_parse(const char* text, zkir_circuit** out);
int32_t zkir_circuit_print(const zkir_circuit* c, char* buf, size_t cap, size_t* needed);

/* CircomTranspiler::transpile_circuit (src/circom.rs:412-429). */
int32_t zkir_circuit_transpile_circom(const zkir_circuit* c, char* buf, size_t cap, size_t* needed);

/* NoirTranspiler::transpile_circuit: the second backend the reference names as a goal
 * (README.md:5,10; no reference code).  Same call shape as the Circom backend
 * (src/circom.rs:4-15,412).  ZKIR_E_UNSUPPORTED when the circuit uses an operator or loop
 * Noir's Field type cannot express; zkir_host_last_error names it. */
int32_t zkir_circuit_transpile_noir(const zkir_circuit* c, char* buf, size_t cap, size_t* needed);

/* `{:#?}` of circuit.instructions as src/main.rs:91 prints it. */
int32_t zkir_circuit_debug(const zkir_circuit* c, char* buf, size_t cap, size_t* needed);

void zkir_circuit_free(zkir_circuit* c);

/* Message of the last failing host-side call (generate / parse / lower ...) on the
 * calling thread; the Rust calls it replaces panic with a message instead
 * (e.g. src/ast.rs:130, src/circom.rs:166-175). */
const char* zkir_host_last_error(void);

/* A Poseidon-style gadget (x^5 S-box, dense mix, round constants) built from
 * ast::Instruction values; README.md:13 lists gadgets as a goal, the reference
 * has no code for them.  width = state elements, rounds = full rounds. */
int32_t zkir_gadget_poseidon_style(uint32_t width, uint32_t rounds, uint64_t seed, zkir_circuit** out);

/* ------------------------------------------------------------------ lower */
/* Outcome classes of lower; mirrors the ways Circom would accept or refuse
 * the transpiled text. */
#define ZKIR_LOWER_OK 0
#define ZKIR_LOWER_DOUBLE_ASSIGN 1   /* a signal constrained twice            */
#define ZKIR_LOWER_OOB_INDEX 2       /* array index out of bounds             */
#define ZKIR_LOWER_DIV_BY_ZERO 3     /* division by a known zero              */
#define ZKIR_LOWER_NON_QUADRATIC 4   /* `<==` right-hand side not quadratic   */
#define ZKIR_LOWER_LOOP_CAP 5        /* unrolling exceeded the op / trip cap  */
#define ZKIR_LOWER_UNKNOWN_CONTROL 6 /* condition or index depends on signals */
#define ZKIR_LOWER_UNINIT_SIGNAL 7   /* signal read before it is assigned     */
#define ZKIR_LOWER_BAD_AST 8         /* malformed tree (unknown id, bad role) */

typedef struct zkir_lower_options {
  uint32_t fold_constants; /* 1: evaluate known sub-expressions on the host   */
  uint32_t n_lslots;       /* on-chip window size to allocate for (0..16)     */
  uint32_t reuse_slots;    /* 1: liveness-based slot reuse; 0: one slot/value */
  uint32_t max_ops;        /* unrolling cap (0 = 1<<22)                       */
  uint32_t bnot_variant;   /* 0: ZKIR_OP_BNOT, 1: ZKIR_OP_BNOT256             */
  uint32_t use_acc;        /* 1: forward results through ZKIR_KIND_ACC        */
  uint32_t max_iterations; /* loop trips over the whole circuit (0 = 1<<20)   */
  uint32_t window_lifetime;/* values read for the last time more than this many
                              ops after their definition go to device memory,
                              not the on-chip window (0 = 48)                 */
  uint32_t strands;        /* 4 or 8: cut the ops into that many strands (a strand
                              tape, for batches too small to fill the machine:
                              8 for the smallest) when the circuit is made of
                              + - * and its cross-strand values fit the exchange
                              slots; 0: one op stream                           */
  uint32_t max_xslots;     /* exchange slots a strand tape may use (0 = 32, or
                              ZKIR_MAX_XSLOTS with 8 strands)                   */
  uint32_t reorder;        /* 1: a plain tape runs the circuit's statements in the
                              order that reads most operands from registers and
                              the window (CHECK ops then carry their constraint
                              numbers, as in strand tapes); 0: program order    */
  uint32_t window_cache;   /* 1: plain tapes keep in the window, at every point,
                              the values read soonest (a value read again later
                              is also copied to a device slot); 0: a value lives
                              in the window or in device memory for all its life */
  uint32_t order_ops;      /* 1: the ops of one statement run in the order that keeps
                              the fewest temporaries alive (`A*B + C`: C first, so the
                              product is forwarded into the sum); 0: as written  */
  uint32_t scaled_values;  /* 1: tapes made of + - * may keep values scaled by 2^-256
                              (ZKIR_OP_INPUT_SCALED / ZKIR_OP_OUTPUT_SCALED) where that
                              saves more output conversions than it adds products;
                              0: every value plain                              */
} zkir_lower_options;

int32_t zkir_lower_options_default(zkir_lower_options* o);

/* Lowerer::new(&circuit).lower().  On ZKIR_E_LOWER *status_class holds one of
 * ZKIR_LOWER_*; opts may be NULL (defaults). */
int32_t zkir_lower(const zkir_circuit* c, const zkir_lower_options* opts, zkir_tape** out, int32_t* status_class);

/* The host side of a campaign in one call (no counterpart in the reference, whose main() makes one
 * circuit: src/main.rs:81-92): for k in 0..n, zkir_generate(seeds[k], &cfgs[k]) then zkir_lower(., opts),
 * spread over `threads` host threads.  tapes[k] receives the tape, or NULL with status_classes[k] set to the
 * ZKIR_LOWER_* class that refused the circuit (that is a result: the call still returns ZKIR_OK).  cfgs may
 * be NULL (defaults for every circuit), opts may be NULL.  On any other error every tape made so far is
 * freed and the code is returned.  The caller frees the tapes with zkir_tape_free.
 * Straight-line configurations (no_loops, no If, max_expr_depth 1, arithmetic_only, quadratic_aware: what a
 * campaign draws) are lowered while they are generated, without the Circuit in between: the same random
 * draws, the same tape, the same refusals, at a third of the host time. */
int32_t zkir_generate_lower_many(const uint64_t* seeds, const zkir_generator_config* cfgs, uint32_t n,
                                 const zkir_lower_options* opts, uint32_t threads, zkir_tape** tapes,
                                 int32_t* status_classes);

/* Thread-seconds zkir_generate_lower_many has spent so far, over all its calls and worker threads of this
 * process: *generate_s in drawing circuits (for circuits lowered while they are drawn: drawing plus the pass that
 * makes single-assignment ops), *lower_s in the rest of lowering (slot placement).  For a campaign's report of
 * where its host time goes; the reference has no timing code (src/main.rs:81-92). */
int32_t zkir_host_phase_seconds(double* generate_s, double* lower_s);

int32_t zkir_tape_header(const zkir_tape* t, zkir_tape_hdr* hdr);
int32_t zkir_tape_blob(const zkir_tape* t, const void** blob, size_t* bytes);
/* Where each signal variable sits in an input (which = 0) or output (which = 1)
 * vector: `triples` receives (variable id, first element, element count) per
 * signal, in the order of Circuit.variables (src/ast.rs:114). */
int32_t zkir_tape_io_map(const zkir_tape* t, int32_t which, uint64_t* triples, size_t cap_entries, size_t* n_entries);
/* wrap an existing header + blob (validates every operand) */
int32_t zkir_tape_from_blob(const zkir_tape_hdr* hdr, const void* blob, size_t bytes, zkir_tape** out);
void zkir_tape_free(zkir_tape* t);

/* ---------------------------------------------------------------- device */
typedef struct zkir_ctx zkir_ctx;
typedef uint32_t zkir_tape_id;

int32_t zkir_ctx_create(int32_t device, zkir_ctx** out);
int32_t zkir_ctx_destroy(zkir_ctx* ctx);
const char* zkir_last_error(const zkir_ctx* ctx);

/* Validate and copy a tape to the device; constants are converted to
 * Montgomery form on the device. */
int32_t zkir_tape_upload(zkir_ctx* ctx, const zkir_tape_hdr* hdr, const void* blob, size_t bytes, zkir_tape_id* out);
/* Give a tape's device memory back.  If a launch that reads the tape may still be running the call waits for
 * THAT launch (the context keeps the events of its last eight launches), not for the device: a caller that works
 * one launch ahead releases the tapes of launch k while launch k + 1 runs.  (After launches made under graph
 * capture or with ZKIR_NO_CHAIN in the environment it waits for the device.) */
int32_t zkir_tape_release(zkir_ctx* ctx, zkir_tape_id id);

/* zkir_tape_upload for n tapes at once: one device allocation, one copy and one
 * conversion kernel for all of them (a campaign uploads hundreds of small tapes per
 * launch; one by one that costs more than evaluating them).  out[0..n) receives the
 * ids; on error no tape of the call is uploaded.  Each id is released on its own. */
int32_t zkir_tape_upload_many(zkir_ctx* ctx, uint32_t n, const zkir_tape_hdr* hdrs, const void* const* blobs,
                              const size_t* bytes, zkir_tape_id* out);

/* zkir_tape_upload_many for tapes the library itself holds (handles from zkir_lower, zkir_generate_lower_many or
 * zkir_tape_from_blob): such a tape was made or validated by the library, so the upload does not look at every
 * op again, which is most of what uploading a campaign's tapes costs. */
int32_t zkir_tape_upload_handles(zkir_ctx* ctx, uint32_t n, const zkir_tape* const* tapes, zkir_tape_id* out);

/* Evaluate one tape for N input vectors; all pointers are HOST memory.
 *   inputs_soa   [n_inputs][8][N]   (may be NULL when n_inputs == 0)
 *   outputs_soa  [n_outputs][8][N]  nullable
 *   unsat_count  number of (vector, constraint) pairs that do not hold
 *   first_unsat  nullable; [2] = {vector, constraint} of the unsatisfied pair
 *                with the smallest vector index (then smallest constraint),
 *                0xffffffff each when every constraint holds
 * N < 2^32; a strand tape takes N < 2^27 per launch and fewer than 466 033 device slots per
 * strand (ZKIR_E_LIMIT beyond). */
int32_t zkir_eval_batch(zkir_ctx* ctx, zkir_tape_id tape, const uint32_t* inputs_soa, uint64_t N,
                        uint32_t* outputs_soa, uint64_t* unsat_count, uint32_t* first_unsat);

/* Same, on DEVICE pointers and a caller stream (hipStream_t as void*, NULL =
 * default stream); asynchronous.  `result_dev` points to 4 uint64 in device
 * memory: {unsat_count, first_unsat packed (vector<<32 | constraint), faults,
 * reserved}; it is initialised by the call. */
int32_t zkir_eval_batch_dev(zkir_ctx* ctx, zkir_tape_id tape, const uint32_t* inputs_soa_dev, uint64_t N,
                            uint32_t* outputs_soa_dev, uint64_t* result_dev, void* stream);

/* zkir_eval_batch_dev for callers that cannot allocate (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same
 * arguments, same results; the work counter and the witness slots live in a fixed array of
 * ZKIR_NOALLOC_SCRATCH_BYTES of device memory that every GPU the library runs on holds from
 * the first zkir_ctx_create until the process ends, and the call makes no hipMalloc / hipFree
 * (the tape and its descriptor are on the device since zkir_tape_upload).  The array has room
 * for ZKIR_NOALLOC_SLOT_ROWS witness-slot rows of 9 KiB: a tape with more device slots than
 * that (n_gslots of its header) is refused with ZKIR_E_LIMIT, and a launch gets only as many
 * workgroups as rows / n_gslots, so big tapes run below the rate of zkir_eval_batch_dev.
 * All contexts of one GPU share the array (two processes do not: each has its own copy): a call that
 * arrives while ANOTHER context's no-alloc launch is still running on that GPU is refused with ZKIR_E_BUSY
 * instead of racing with it; calls of one context are chained like every other launch of it.
 * Under graph capture (the stream is being recorded into a HIP graph) the call records its two kernels (one
 * that initialises the result words and the work counter, then the interpreter) and nothing else: it neither waits for nor records the context's events, and the busy check is
 * skipped.  The caller then synchronises with its tape uploads before the capture begins, and keeps replays
 * of the graph from overlapping any other launch that uses the same context or the same GPU's array. */
#define ZKIR_NOALLOC_SCRATCH_BYTES (64u << 20)
#define ZKIR_NOALLOC_SLOT_ROWS (ZKIR_NOALLOC_SCRATCH_BYTES / 9216u - 1u)
int32_t zkir_eval_batch_dev_noalloc(zkir_ctx* ctx, zkir_tape_id tape, const uint32_t* inputs_soa_dev, uint64_t N,
                                    uint32_t* outputs_soa_dev, uint64_t* result_dev, void* stream);

/* Many tapes, N_per_tape vectors each: one tape per workgroup.
 *   inputs  concatenation of each tape's [n_inputs][8][N_per_tape] block
 *   per_tape_unsat [n_tapes]  unsatisfied pairs per tape
 * Host pointers.  Plain and strand tapes may share a call (they run as two launches,
 * since the two kinds use a workgroup differently). */
int32_t zkir_eval_multi(zkir_ctx* ctx, const zkir_tape_id* tapes, uint32_t n_tapes, const uint32_t* inputs,
                        uint64_t N_per_tape, uint64_t* per_tape_unsat);

/* Device-pointer form of zkir_eval_multi; per_tape_unsat_dev [n_tapes] uint64.  result_dev
 * (nullable) points to 4 uint64 in device memory, initialised by the call:
 * {0, smallest (vector<<32 | constraint) over the unsatisfied pairs of all tapes or all ones,
 *  faults (vectors that divided by zero at run time, over all tapes), reserved}. */
int32_t zkir_eval_multi_dev(zkir_ctx* ctx, const zkir_tape_id* tapes, uint32_t n_tapes, const uint32_t* inputs_dev,
                            uint64_t N_per_tape, uint64_t* per_tape_unsat_dev, uint64_t* result_dev, void* stream);

/* zkir_eval_multi_dev that also says WHICH tapes faulted: per_tape_faults_dev [n_tapes] uint64 (nullable; with
 * NULL this is zkir_eval_multi_dev) receives, per tape, the vectors that divided by zero at run time, so a
 * campaign can name the circuit that faulted and not just the launch.  No counterpart in the reference. */
int32_t zkir_eval_multi_dev_faults(zkir_ctx* ctx, const zkir_tape_id* tapes, uint32_t n_tapes, const uint32_t* inputs_dev,
                                   uint64_t N_per_tape, uint64_t* per_tape_unsat_dev, uint64_t* per_tape_faults_dev,
                                   uint64_t* result_dev, void* stream);

/* ------------------------------------------------------- several GPUs, one process */
/* The campaign's one exchange step for a host process that drives several GPUs itself (one zkir_ctx per device,
 * circuit i on device i mod n: units are independent, nothing else is exchanged): RCCL communicators over the given
 * devices (ncclCommInitAll), one all-reduce (sum) of `n_counters` uint64 counters per device and one all-gather of
 * each device's list of divergent circuit ids (at most `id_capacity` each; the count travels first, the payload is
 * padded).  No counterpart in the reference, whose only caller is one process with one backend (src/main.rs:81-92).
 *   counters [n_devices][n_counters], ids [n_devices][id_capacity], n_ids [n_devices]          (host memory, in)
 *   totals [n_counters], all_ids [n_devices * id_capacity] sorted without repeats, *n_all_ids   (host memory, out)
 * ZKIR_E_UNSUPPORTED when no RCCL library can be loaded, ZKIR_E_LIMIT when a device brings more ids than the
 * capacity.  A process per GPU (torch.distributed, MPI) does the same exchange with its own collectives. */
typedef struct zkir_reducer zkir_reducer;
int32_t zkir_reducer_create(const int32_t* devices, uint32_t n_devices, uint32_t n_counters, uint32_t id_capacity, zkir_reducer** out);
int32_t zkir_reducer_reduce(zkir_reducer* r, const uint64_t* counters, const uint32_t* ids, const uint32_t* n_ids, uint64_t* totals,
                            uint32_t* all_ids, uint32_t* n_all_ids);
const char* zkir_reducer_last_error(const zkir_reducer* r);
int32_t zkir_reducer_destroy(zkir_reducer* r);

/* Faults (division by zero at run time) seen by the last synchronous call. */
int32_t zkir_last_faults(const zkir_ctx* ctx, uint64_t* faults);

/* Measured issue rate of one integer-multiply instruction on this device, in
 * lane-operations per second: which = 0 v_mul_hi_u32, 1 v_mul_lo_u32,
 * 2 v_mad_u64_u32.  Used as the denominator of the multiply roofline. */
int32_t zkir_measure_int_mul_rate(zkir_ctx* ctx, int32_t which, double* lane_ops_per_s);

/* Raw Montgomery multiplications per second of the field multiplier alone
 * (registers only, no tape): the ceiling of any tape made of ZKIR_OP_MUL. */
int32_t zkir_measure_fr_mul_rate(zkir_ctx* ctx, double* fr_mul_per_s);

/* The same for the multiplier on 9 limbs of 29 bits (Montgomery radix 2^261) that the interpreter
 * uses for tapes whose values stay on chip. */
int32_t zkir_measure_fr_mul_rate29(zkir_ctx* ctx, double* fr_mul_per_s);

#ifdef __cplusplus
}
#endif
#endif /* ZKIR_HIP_H */
