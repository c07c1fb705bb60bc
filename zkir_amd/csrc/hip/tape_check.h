// Structural validation of a tape: every index an op names must lie inside the
// tables the header declares, so that no kernel can be launched on a tape that
// would read or write outside its buffers.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../../include/zkir_hip.h"

namespace zkir {

inline bool check_source(const zkir_tape_hdr& h, uint32_t o, std::string* why) {
  const uint32_t kind = o >> ZKIR_KIND_SHIFT, idx = o & ZKIR_INDEX_MASK;
  switch (kind) {
    case ZKIR_KIND_GSLOT: if (idx < h.n_gslots) return true; break;
    case ZKIR_KIND_LSLOT: if (idx < h.n_lslots) return true; break;
    case ZKIR_KIND_CONST: if (idx < h.n_consts) return true; break;
    case ZKIR_KIND_ACC: return true;
    case ZKIR_KIND_XSLOT: if ((h.flags & ZKIR_TAPE_F_STRANDS) && idx < h.n_xslots) return true; break;
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
  if (kind == ZKIR_KIND_XSLOT && (h.flags & ZKIR_TAPE_F_STRANDS) && idx < h.n_xslots) return true;
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
  if (h.flags & ~ZKIR_TAPE_F_STRANDS) return bad("unknown flag bits");
  const bool strands = (h.flags & ZKIR_TAPE_F_STRANDS) != 0;
  if (h.n_scale_ops > h.n_mul_ops) return bad("more scaling products than products");
  uint32_t any_strand_ops = 0;
  for (uint32_t s = 0; s < ZKIR_MAX_STRANDS; s++) any_strand_ops |= h.strand_ops[s];
  if (strands) {
    if (h.n_strands != ZKIR_STRANDS && h.n_strands != ZKIR_MAX_STRANDS) return bad("a strand tape has 4 or 8 strands");
    uint64_t sum = 0;
    for (uint32_t s = 0; s < ZKIR_MAX_STRANDS; s++) {
      if (s >= h.n_strands && h.strand_ops[s]) return bad("op count set for a strand the tape does not have");
      sum += h.strand_ops[s];
    }
    if (sum != h.n_ops) return bad("strand op counts do not add up to n_ops");
    if (h.n_xslots > ZKIR_MAX_XSLOTS) return bad("too many exchange slots");
  } else if (any_strand_ops | h.n_xslots | h.n_strands) {
    return bad("strand fields set on a plain tape");
  }
  if (h.n_ops > (1u << 26) || h.n_consts > (1u << 26) || h.n_gslots > (1u << 24) || h.n_inputs > (1u << 24) ||
      h.n_outputs > (1u << 24))
    return bad("table size beyond limit");
  const size_t need = (size_t)h.n_ops * sizeof(zkir_op) + (size_t)h.n_consts * 32 + (size_t)h.n_constraints * 12;
  if (bytes != need) return bad("blob size does not match header");
  if (need && !blob) return bad("null blob");
  const zkir_op* ops = (const zkir_op*)blob;
  uint32_t checks = 0;
  // strand tapes: barriers per strand (all strands must hold the same number, or the
  // wavefronts of a workgroup would wait for each other forever), and every constraint
  // number exactly once
  uint32_t barriers[ZKIR_MAX_STRANDS] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t strand = 0, strand_end = strands ? h.strand_ops[0] : h.n_ops;
  // CHECK ops carry their constraint number + 1 in bits 8..31 of `code`: all of them in a strand tape, all or none
  // of them in a plain tape (none: the k-th CHECK op is constraint k)
  std::vector<bool> seen_check;
  uint32_t numbered = 0;
  for (uint32_t i = 0; i < h.n_ops; i++) {
    const zkir_op& op = ops[i];
    while (strands && i >= strand_end && strand + 1 < h.n_strands) strand_end += h.strand_ops[++strand];
    const uint32_t code = op.code & 0xffu;
    if (op.code & ~0xffu) {
      if (code != ZKIR_OP_CHECK) return bad("reserved opcode bits set");
      if (seen_check.empty()) seen_check.assign(h.n_constraints, false);
      const uint32_t id = (op.code >> 8) - 1;
      if (id >= h.n_constraints || seen_check[id]) return bad("constraint number out of range or repeated");
      seen_check[id] = true;
      numbered++;
    } else if (strands && code == ZKIR_OP_CHECK) {
      return bad("CHECK op of a strand tape carries no constraint number");
    }
    if (code == ZKIR_OP_NOP) continue;
    if (code == ZKIR_OP_BARRIER) {
      if (!strands) return bad("BARRIER in a plain tape");
      barriers[strand]++;
      continue;
    }
    if (code == ZKIR_OP_INPUT || code == ZKIR_OP_INPUT_SCALED) {
      if ((op.a >> ZKIR_KIND_SHIFT) != ZKIR_KIND_INPUT || (op.a & ZKIR_INDEX_MASK) >= h.n_inputs)
        return bad("input index out of range");
      if (!check_dest(h, op.dst, why)) return false;
    } else if (code == ZKIR_OP_OUTPUT || code == ZKIR_OP_OUTPUT_SCALED) {
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
  if (numbered && numbered != checks) return bad("some CHECK ops carry a constraint number and some do not");
  for (uint32_t s = 1; strands && s < h.n_strands; s++)
    if (barriers[s] != barriers[0]) return bad("strands hold different numbers of barriers");
  return true;
}

}  // namespace zkir
