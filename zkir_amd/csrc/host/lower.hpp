// Circuit -> Tape.  A second backend beside the Circom printer: like
// CircomTranspiler (reference src/circom.rs:4-15) it borrows a `Circuit` and has
// one method.  The reference has nothing to flatten an AST; what each statement
// means is taken from the Circom text the reference prints for it
// (src/circom.rs:161-410).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/zkir_hip.h"
#include "ast.hpp"

namespace zkir {

struct Tape {
  zkir_tape_hdr hdr{};
  std::vector<uint8_t> blob;  // ops | consts | triples, as include/zkir_hip.h lays them out
  // where each signal variable sits in an input / output vector
  struct IoEntry {
    size_t var_id, offset, len;
  };
  std::vector<IoEntry> inputs, outputs;
};

struct LowerError : std::runtime_error {
  int32_t status_class;  // ZKIR_LOWER_*
  LowerError(int32_t c, const std::string& m) : std::runtime_error(m), status_class(c) {}
};

using LowerOptions = zkir_lower_options;
LowerOptions default_lower_options();

class Lowerer {
 public:
  explicit Lowerer(const Circuit& circuit) : circuit(circuit), opts_(default_lower_options()) {}
  Lowerer(const Circuit& circuit, const LowerOptions& o) : circuit(circuit), opts_(o) {}
  Tape lower();  // throws LowerError
  const Circuit& circuit;

 private:
  LowerOptions opts_;
};

// Generator(Rng::from_seed(seed), cfg).generate() followed by Lowerer(circuit, opts).lower() without the circuit in
// between, for the configurations where that is possible (can_lower_generated: straight-line, quadratic-aware
// campaign circuits): same random draws, same tape, same refusals.  pass1_seconds (nullable) receives the time up
// to the single-assignment ops (drawing the circuit and pass 1); the rest of the call is slot placement.
bool can_lower_generated(const zkir_generator_config& cfg, const LowerOptions& opts);
Tape lower_generated(uint64_t seed, const zkir_generator_config& cfg, const LowerOptions& opts, double* pass1_seconds = nullptr);

}  // namespace zkir

struct zkir_tape {
  zkir::Tape tape;
};
