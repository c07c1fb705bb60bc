// Structures shared by the host runtime and the gfx950 kernels.
#pragma once
#include <stdint.h>

#include "../../../include/zkir_hip.h"

#define ZKIR_BLOCK_THREADS 256

namespace zkir {

// Device-side description of one uploaded tape (64 bytes, read with scalar loads).
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
  uint32_t reserved[3];
};
static_assert(sizeof(TapeDesc) == 64, "TapeDesc must stay 64 bytes");

}  // namespace zkir
