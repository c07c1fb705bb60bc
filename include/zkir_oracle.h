/*
 * zkir_oracle.h - the scalar CPU interpreter of tapes: TEST INFRASTRUCTURE.
 *
 * zkir_eval_cpu is the CPU counterpart of zkir_eval_batch (include/zkir_hip.h): same tape, same buffer
 * layout, same meaning of the results, plus a thread count.  It is what the GPU kernels are checked against
 * and what bench.py times as the CPU baseline.  The reference has no evaluator to mirror (`Field` is a
 * `String`, src/ast.rs:12); the arithmetic meaning of each opcode is the Circom operator src/circom.rs:35-157
 * prints for it.
 *
 * The symbol is exported by oracle/_build/libzkir_oracle.so (built from oracle/eval_oracle.c by
 * oracle/Makefile), never by libzkir_hip.so: the product has no CPU path.
 */
#ifndef ZKIR_ORACLE_H
#define ZKIR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* hdr: a zkir_tape_hdr; blob: its ops | consts | triples (the tape is trusted: build it with zkir_lower or
 * validate it with zkir_tape_from_blob first).  inputs_soa [n_inputs][8][N], outputs_soa [n_outputs][8][N]
 * nullable, first_unsat[2] = {vector, constraint} or 0xffffffff each, faults = vectors that divided by zero.
 * The vectors are split into contiguous ranges over n_threads host threads (1..1024).
 * Returns 0, or -1 on a bad argument. */
int zkir_eval_cpu(const void* hdr, const void* blob, const uint32_t* inputs_soa, uint64_t N, uint32_t* outputs_soa,
                  uint64_t* unsat_count, uint32_t* first_unsat, uint64_t* faults, int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* ZKIR_ORACLE_H */
