// The zkIR data model, one to one with the reference's src/ast.rs: the same
// type names, variants and fields, so code written against the Rust crate reads
// the same here.  Every backend (the Circom printer, the lowerer) consumes these.
#pragma once
#include <cstddef>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace zkir {

// src/ast.rs:3-10
struct VarRef {
  size_t id = 0;                 // unique id of the variable
  std::optional<size_t> idx;     // element of an array; arrays are never referenced whole
};

// src/ast.rs:12: a field element is carried as its decimal text
using Field = std::string;

// src/ast.rs:14-42; Not is commented out in the reference (ast.rs:35) and has no variant here
enum class ExprKind {
  Var, Constant,
  Add, Sub, Mul, Power, Div, IntDiv, Rem,
  LessThan, LessThanEq, GreaterThan, GreaterThanEq, Equal,
  And, Or,
  BitAnd, BitOr, BitNot, BitXor, BitRShift, BitLShift,
};

struct Expr {
  ExprKind kind = ExprKind::Constant;
  VarRef var;                     // Var
  Field constant;                 // Constant
  std::shared_ptr<Expr> left;     // binary ops, and the operand of BitNot
  std::shared_ptr<Expr> right;    // binary ops

  static Expr Var(VarRef v) { Expr e; e.kind = ExprKind::Var; e.var = std::move(v); return e; }
  static Expr Constant(Field f) { Expr e; e.kind = ExprKind::Constant; e.constant = std::move(f); return e; }
  static Expr Binary(ExprKind k, Expr l, Expr r) {
    Expr e; e.kind = k;
    e.left = std::make_shared<Expr>(std::move(l));
    e.right = std::make_shared<Expr>(std::move(r));
    return e;
  }
  static Expr Unary(ExprKind k, Expr x) {
    Expr e; e.kind = k;
    e.left = std::make_shared<Expr>(std::move(x));
    return e;
  }
  bool is_binary() const { return kind != ExprKind::Var && kind != ExprKind::Constant && kind != ExprKind::BitNot; }
};

// src/ast.rs:44-48
struct VariableType {
  bool is_array = false;
  size_t len = 0;  // Array(len)
  static VariableType Field() { return {}; }
  static VariableType Array(size_t n) { VariableType t; t.is_array = true; t.len = n; return t; }
  bool operator==(const VariableType& o) const { return is_array == o.is_array && (!is_array || len == o.len); }
};

// src/ast.rs:50-60
enum class SignalType { Input, Output };
struct VariableRole {
  bool is_signal = false;
  SignalType signal = SignalType::Input;
  static VariableRole Local() { return {}; }
  static VariableRole Signal(SignalType s) { VariableRole r; r.is_signal = true; r.signal = s; return r; }
  bool is_local() const { return !is_signal; }
  bool is_input() const { return is_signal && signal == SignalType::Input; }
  bool is_output() const { return is_signal && signal == SignalType::Output; }
  bool operator==(const VariableRole& o) const { return is_signal == o.is_signal && (!is_signal || signal == o.signal); }
};

// src/ast.rs:62-88
struct Variable {
  VarRef var_ref;
  std::string name;
  VariableType var_type;
  VariableRole role;
  std::vector<bool> initialized;

  bool all_initialized() const {
    for (bool b : initialized) if (!b) return false;
    return true;
  }
  bool any_initialized() const {
    for (bool b : initialized) if (b) return true;
    return false;
  }
  std::vector<size_t> get_initialized() const {
    std::vector<size_t> r;
    for (size_t i = 0; i < initialized.size(); i++) if (initialized[i]) r.push_back(i);
    return r;
  }
};

// src/ast.rs:90-111
enum class InstrKind { VarDecl, Assign, ArrayAssign, Constraint, If, For, While };

struct Instruction {
  InstrKind kind = InstrKind::VarDecl;
  Variable var;                          // VarDecl
  std::optional<Expr> default_value;     // VarDecl
  VarRef target;                         // Assign / ArrayAssign / Constraint
  std::optional<Expr> index;             // ArrayAssign: the index expression
  std::optional<Expr> value;             // Assign / ArrayAssign / Constraint right-hand side
  std::optional<Expr> cond;              // If / For / While
  std::vector<Instruction> then_branch;  // If
  std::optional<std::vector<Instruction>> else_branch;  // If
  std::shared_ptr<Instruction> init;     // For
  std::shared_ptr<Instruction> step;     // For
  std::vector<Instruction> body;         // For / While

  static Instruction VarDecl(Variable v, std::optional<Expr> def) {
    Instruction i; i.kind = InstrKind::VarDecl; i.var = std::move(v); i.default_value = std::move(def); return i;
  }
  static Instruction Assign(VarRef t, Expr e) {
    Instruction i; i.kind = InstrKind::Assign; i.target = std::move(t); i.value = std::move(e); return i;
  }
  static Instruction ArrayAssign(VarRef t, Expr idx, Expr e) {
    Instruction i; i.kind = InstrKind::ArrayAssign; i.target = std::move(t); i.index = std::move(idx);
    i.value = std::move(e); return i;
  }
  static Instruction Constraint(VarRef t, Expr e) {
    Instruction i; i.kind = InstrKind::Constraint; i.target = std::move(t); i.value = std::move(e); return i;
  }
  static Instruction If(Expr c, std::vector<Instruction> th, std::optional<std::vector<Instruction>> el) {
    Instruction i; i.kind = InstrKind::If; i.cond = std::move(c); i.then_branch = std::move(th);
    i.else_branch = std::move(el); return i;
  }
  static Instruction For(Instruction init, Expr c, Instruction step, std::vector<Instruction> body) {
    Instruction i; i.kind = InstrKind::For;
    i.init = std::make_shared<Instruction>(std::move(init));
    i.cond = std::move(c);
    i.step = std::make_shared<Instruction>(std::move(step));
    i.body = std::move(body); return i;
  }
  static Instruction While(Expr c, std::vector<Instruction> body) {
    Instruction i; i.kind = InstrKind::While; i.cond = std::move(c); i.body = std::move(body); return i;
  }
};

// src/ast.rs:113-132
struct Circuit {
  std::vector<Variable> variables;
  std::vector<Instruction> instructions;

  Circuit() = default;
  Circuit(std::vector<Variable> v, std::vector<Instruction> i) : variables(std::move(v)), instructions(std::move(i)) {}

  // linear scan, as in the reference (ast.rs:126-131); throws where the Rust panics
  const Variable& get_variable(const VarRef& r) const {
    for (const Variable& v : variables)
      if (v.var_ref.id == r.id) return v;
    throw std::runtime_error("Variable " + std::to_string(r.id) + " not found");
  }
};

}  // namespace zkir

// opaque handle of the C ABI
struct zkir_circuit {
  zkir::Circuit circuit;
  size_t requested = 0, generated = 0;  // what Generator::generate reported, if it made this circuit
};
