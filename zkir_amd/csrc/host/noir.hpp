// AST -> Noir source text: the second backend the reference's README names as a goal
// (/root/reference/README.md:5,10) and has no code for.  Same shape as the Circom backend
// (a struct borrowing the circuit, one transpile_circuit method; /root/reference/src/circom.rs:4-15,412).
//
// Noir is not a signal language: inputs become the parameters of `main`, output signals
// become mutable locals returned as `pub` values, `out <== e` becomes `out = e` (computing a
// returned value is what constrains it).  Only what Noir's `Field` type can express is
// printed: + - * / and == on expressions, `if` on "expression != 0", `for` loops of the
// shape the generator emits (constant start, `v <= constant`, `v = v + 1`).  Everything else
// (integer division, ordering, bit operations, `**`, `while`, writes to a loop variable)
// throws NoirUnsupported naming the construct: those operators have no Field counterpart in
// Noir, and mapping them through integer casts would change their meaning on 254-bit values.
// No Noir compiler exists in the build environment: the syntax printed here has not met one.
#pragma once
#include <set>
#include <stdexcept>
#include <string>

#include "ast.hpp"

namespace zkir {

struct NoirUnsupported : std::runtime_error {
  using std::runtime_error::runtime_error;
};

class NoirTranspiler {
 public:
  explicit NoirTranspiler(const Circuit& circuit) : circuit(circuit), indent_size_(1) {}
  std::string transpile_circuit();
  const Circuit& circuit;

 private:
  std::string get_indent() const;
  std::string type_of(const Variable& v) const;
  std::string transpile_expr(const Expr& e) const;
  std::string transpile_instruction(const Instruction& instr);
  std::string transpile_block(const std::vector<Instruction>& body);
  size_t indent_size_;
  std::set<size_t> loop_vars_;  // ids of the loop variables of the enclosing `for` loops
};

}  // namespace zkir
