// Text form of a Circuit used at the C ABI: an s-expression that carries exactly
// the fields of the reference's src/ast.rs types.
//
//   (circuit
//     (vars (var <id> "<name>" field|(array <n>) local|input|output (init <0|1>...)) ...)
//     (instrs <instr> ...))
//   <instr> := (decl <id> [<expr>])            VarDecl(variable <id>, default)
//            | (assign <ref> <expr>) | (aassign <ref> <expr-index> <expr>)
//            | (constrain <ref> <expr>)
//            | (if <expr> (then <instr>...) [(else <instr>...)])
//            | (for <instr-init> <expr-cond> <instr-step> (body <instr>...))
//            | (while <expr> (body <instr>...))
//   <ref>   := <id> | (<id> <idx>)
//   <expr>  := (const "<decimal>") | (var <ref>) | (<op> <expr> <expr>) | (bnot <expr>)
//   <op>    := add sub mul pow div idiv rem lt le gt ge eq and or band bor bxor shr shl
#pragma once
#include <string>

#include "ast.hpp"

namespace zkir {

std::string print_circuit(const Circuit& c);
// throws std::runtime_error with a message on malformed text
Circuit parse_circuit(const std::string& text);

}  // namespace zkir
