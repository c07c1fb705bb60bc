// See gadget.hpp.  The reference lists "gadgets shared among all zkDSLs" as a goal
// (README.md:13) and has no code for them; this builder only uses the reference's
// own AST vocabulary (src/ast.rs:91-111), so the result prints through the Circom
// backend like any generated circuit.
#include "gadget.hpp"

#include <stdexcept>
#include <string>

#include "rng.hpp"

namespace zkir {

Circuit poseidon_style_gadget(uint32_t width, uint32_t rounds, uint64_t seed) {
  if (width == 0 || width > 16 || rounds == 0) throw std::invalid_argument("gadget: width 1..16, rounds >= 1");
  Rng rng = Rng::from_seed(seed);
  Circuit c;
  size_t next_id = 0;
  auto add_var = [&](const std::string& name, VariableRole role) {
    Variable v;
    v.var_ref.id = next_id++;
    v.name = name;
    v.var_type = VariableType::Field();
    v.role = role;
    v.initialized.assign(1, false);
    c.variables.push_back(v);
    return v;
  };
  Arena& ar = *c.arena;
  auto cst = [&]() { return Expr::Constant(rng.next()); };
  auto var = [](const Variable& v) { return Expr::Var(v.var_ref); };

  std::vector<Expr> mix;  // width x width constants, reused by every round
  for (uint32_t i = 0; i < width * width; i++) mix.push_back(cst());

  std::vector<Variable> state;
  for (uint32_t i = 0; i < width; i++) {
    Variable in = add_var("in" + std::to_string(i), VariableRole::Signal(SignalType::Input));
    c.instructions.push_back(Instruction::VarDecl(ar, in, std::nullopt));
    Variable s = add_var("s" + std::to_string(i), VariableRole::Local());
    c.instructions.push_back(Instruction::VarDecl(ar, s, var(in)));
    state.push_back(s);
  }
  for (uint32_t r = 0; r < rounds; r++) {
    const std::string tag = "_" + std::to_string(r) + "_";
    std::vector<Variable> sboxed;
    for (uint32_t i = 0; i < width; i++) {
      const std::string t = tag + std::to_string(i);
      // a = s + round constant
      Variable a = add_var("a" + t, VariableRole::Local());
      c.instructions.push_back(Instruction::VarDecl(ar, a, Expr::Binary(ar, ExprKind::Add, var(state[i]), cst())));
      // x^5 as three rank-1 constraints: sq = a*a, qd = sq*sq, ft = qd*a
      Variable sq = add_var("sq" + t, VariableRole::Signal(SignalType::Output));
      c.instructions.push_back(Instruction::VarDecl(ar, sq, std::nullopt));
      c.instructions.push_back(Instruction::Constraint(ar, sq.var_ref, Expr::Binary(ar, ExprKind::Mul, var(a), var(a))));
      Variable qd = add_var("qd" + t, VariableRole::Signal(SignalType::Output));
      c.instructions.push_back(Instruction::VarDecl(ar, qd, std::nullopt));
      c.instructions.push_back(Instruction::Constraint(ar, qd.var_ref, Expr::Binary(ar, ExprKind::Mul, var(sq), var(sq))));
      Variable ft = add_var("ft" + t, VariableRole::Signal(SignalType::Output));
      c.instructions.push_back(Instruction::VarDecl(ar, ft, std::nullopt));
      c.instructions.push_back(Instruction::Constraint(ar, ft.var_ref, Expr::Binary(ar, ExprKind::Mul, var(qd), var(a))));
      sboxed.push_back(ft);
    }
    // dense mix: s_i = sum_j M[i][j] * ft_j
    std::vector<Variable> next;
    for (uint32_t i = 0; i < width; i++) {
      Expr acc = Expr::Binary(ar, ExprKind::Mul, mix[i * width], var(sboxed[0]));
      for (uint32_t j = 1; j < width; j++)
        acc = Expr::Binary(ar, ExprKind::Add, acc, Expr::Binary(ar, ExprKind::Mul, mix[i * width + j], var(sboxed[j])));
      Variable s = add_var("m" + tag + std::to_string(i), VariableRole::Local());
      c.instructions.push_back(Instruction::VarDecl(ar, s, acc));
      next.push_back(s);
    }
    state = next;
  }
  return c;
}

}  // namespace zkir
