// Structural validation of a tape: every index an op names must lie inside the
// tables the header declares, so that no kernel can be launched on a tape that
// would read or write outside its buffers.
#pragma once
#include <stdint.h>

#include <string>

#include "../../../include/zkir_hip.h"

namespace zkir {

inline bool check_source(const zkir_tape_hdr& h, uint32_t o, std::string* why) {
  const uint32_t kind = o >> ZKIR_KIND_SHIFT, idx = o & ZKIR_INDEX_MASK;
  switch (kind) {
    case ZKIR_KIND_GSLOT: if (idx < h.n_gslots) return true; break;
    case ZKIR_KIND_LSLOT: if (idx < h.n_lslots) return true; break;
    case ZKIR_KIND_CONST: if (idx < h.n_consts) return true; break;
    case ZKIR_KIND_ACC: return true;
    default: break;
  }
  if (why) *why = "source operand out of range";
  return false;
}

inline bool check_dest(const zkir_tape_hdr& h, uint32_t o, std::string* why) {
  const uint32_t kind = o >> ZKIR_KIND_SHIFT, idx = o & ZKIR_INDEX_MASK;
  if (kind == ZKIR_KIND_NONE) return true;
  if (kind == ZKIR_KIND_GSLOT && idx < h.n_gslots) return true;
  if (kind == ZKIR_KIND_LSLOT && idx < h.n_lslots) return true;
  if (why) *why = "destination operand out of range";
  return false;
}

inline bool check_tape(const zkir_tape_hdr& h, const void* blob, size_t bytes, std::string* why) {
  auto bad = [&](const char* m) {
    if (why) *why = m;
    return false;
  };
  if (h.magic != ZKIR_TAPE_MAGIC || h.version != ZKIR_TAPE_VERSION) return bad("magic/version");
  if (h.n_lslots > ZKIR_MAX_LSLOTS) return bad("too many on-chip slots");
  if (h.n_ops > (1u << 26) || h.n_consts > (1u << 26) || h.n_gslots > (1u << 24) || h.n_inputs > (1u << 24) ||
      h.n_outputs > (1u << 24))
    return bad("table size beyond limit");
  const size_t need = (size_t)h.n_ops * sizeof(zkir_op) + (size_t)h.n_consts * 32 + (size_t)h.n_constraints * 12;
  if (bytes != need) return bad("blob size does not match header");
  if (need && !blob) return bad("null blob");
  const zkir_op* ops = (const zkir_op*)blob;
  uint32_t checks = 0;
  for (uint32_t i = 0; i < h.n_ops; i++) {
    const zkir_op& op = ops[i];
    if (op.code & ~0xffu) return bad("reserved opcode bits set");
    const uint32_t code = op.code;
    if (code == ZKIR_OP_NOP) continue;
    if (code == ZKIR_OP_INPUT) {
      if ((op.a >> ZKIR_KIND_SHIFT) != ZKIR_KIND_INPUT || (op.a & ZKIR_INDEX_MASK) >= h.n_inputs)
        return bad("input index out of range");
      if (!check_dest(h, op.dst, why)) return false;
    } else if (code == ZKIR_OP_OUTPUT) {
      if (op.dst >= h.n_outputs) return bad("output index out of range");
      if (!check_source(h, op.a, why)) return false;
    } else if (code == ZKIR_OP_CHECK) {
      if (!check_source(h, op.a, why) || !check_source(h, op.b, why) || !check_source(h, op.dst, why)) return false;
      checks++;
    } else if (code >= ZKIR_OP_COPY && code <= ZKIR_OP_BNOT256) {
      if (!check_source(h, op.a, why) || !check_dest(h, op.dst, why)) return false;
    } else if (code >= ZKIR_OP_ADD && code <= ZKIR_OP_SHL) {
      if (!check_source(h, op.a, why) || !check_source(h, op.b, why) || !check_dest(h, op.dst, why)) return false;
    } else {
      return bad("unknown opcode");
    }
  }
  if (checks != h.n_constraints) return bad("n_constraints does not match the CHECK ops");
  return true;
}

}  // namespace zkir
