// AST -> Circom source text: the reference's src/circom.rs, byte for byte.
#pragma once
#include <string>

#include "ast.hpp"

namespace zkir {

class CircomTranspiler {
 public:
  explicit CircomTranspiler(const Circuit& circuit) : circuit(circuit), indent_size_(1) {}  // circom.rs:10-15
  std::string transpile_circuit();                                                           // circom.rs:412-429
  const Circuit& circuit;

 private:
  std::string get_indent() const;                               // circom.rs:17-19
  std::string transpile_expr(const Expr& e) const;              // circom.rs:21-159
  std::string transpile_instruction(const Instruction& instr);  // circom.rs:161-410
  std::string transpile_block(const std::vector<Instruction>& body);
  size_t indent_size_;
};

// `{:#?}` of a Vec<Instruction>, what src/main.rs:91 prints
std::string debug_instructions(const std::vector<Instruction>& instrs);

}  // namespace zkir
