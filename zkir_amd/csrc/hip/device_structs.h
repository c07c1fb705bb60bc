// Structures shared by the host runtime and the gfx950 kernels.
#pragma once
#include <stdint.h>

#include "../../../include/zkir_hip.h"

#define ZKIR_BLOCK_THREADS 256

namespace zkir {

// Device-side description of one uploaded tape (144 bytes, read with scalar loads).
struct TapeDesc {
  const zkir_op* ops;      // n_ops entries
  const uint32_t* consts;  // n_consts x 8 limbs, Montgomery form
  uint64_t in_word_off;    // offset (in uint32) of this tape's block in the inputs buffer
  uint64_t out_word_off;   // same for outputs
  uint32_t n_ops;
  uint32_t n_inputs;
  uint32_t n_outputs;
  uint32_t n_constraints;
  uint32_t n_consts;
  uint32_t n_xslots;       // strand tapes: exchange slots
  uint32_t unsat_index;    // which entry of the launch's per-tape unsat array this tape counts into
  uint32_t n_strands;      // strand tapes: 4 or 8
  uint32_t strand_begin[ZKIR_MAX_STRANDS];  // strand tapes: first op of each strand
  uint32_t strand_ops[ZKIR_MAX_STRANDS];    // strand tapes: ops of each strand
  const uint32_t* consts29;             // n_consts x 12 words: 9 limbs of 29 bits, Montgomery form with R = 2^261
  uint32_t one_word;       // operand word of the constant 1 in this tape's pool (0xffffffff: none): a CHECK whose
                           // right factor is that word compares its left factor as it is
  uint32_t reserved;
};
static_assert(sizeof(TapeDesc) == 144, "TapeDesc layout");

}  // namespace zkir
