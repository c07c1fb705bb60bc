// See noir.hpp.
#include "noir.hpp"

namespace zkir {

std::string NoirTranspiler::get_indent() const { return std::string(indent_size_ * 4, ' '); }

std::string NoirTranspiler::type_of(const Variable& v) const {
  return v.var_type.is_array ? "[Field; " + std::to_string(v.var_type.len) + "]" : "Field";
}

static const char* arith_token(ExprKind k) {
  switch (k) {
    case ExprKind::Add: return " + ";
    case ExprKind::Sub: return " - ";
    case ExprKind::Mul: return " * ";
    case ExprKind::Div: return " / ";
    default: return nullptr;
  }
}

static const char* kind_name(ExprKind k) {
  switch (k) {
    case ExprKind::Power: return "**";
    case ExprKind::IntDiv: return "\\";
    case ExprKind::Rem: return "%";
    case ExprKind::LessThan: return "<";
    case ExprKind::LessThanEq: return "<=";
    case ExprKind::GreaterThan: return ">";
    case ExprKind::GreaterThanEq: return ">=";
    case ExprKind::And: return "&&";
    case ExprKind::Or: return "||";
    case ExprKind::BitAnd: return "&";
    case ExprKind::BitOr: return "|";
    case ExprKind::BitNot: return "~";
    case ExprKind::BitXor: return "^";
    case ExprKind::BitRShift: return ">>";
    case ExprKind::BitLShift: return "<<";
    default: return "?";
  }
}

std::string NoirTranspiler::transpile_expr(const Expr& e) const {
  switch (e.kind) {
    case ExprKind::Var: {
      const Variable& var = circuit.get_variable(e.var);
      if (e.var.idx) return var.name + "[" + std::to_string(*e.var.idx) + "]";
      return var.name;
    }
    case ExprKind::Constant:
      return e.constant.str();
    case ExprKind::Equal:
      return "((" + transpile_expr(*e.left) + " == " + transpile_expr(*e.right) + ") as Field)";
    default:
      if (const char* tok = arith_token(e.kind))
        return "(" + transpile_expr(*e.left) + tok + transpile_expr(*e.right) + ")";
      throw NoirUnsupported(std::string("operator ") + kind_name(e.kind) + " has no Field counterpart in Noir");
  }
}

std::string NoirTranspiler::transpile_block(const std::vector<Instruction>& body) {
  std::string s;
  for (const Instruction& i : body) {
    const std::string line = transpile_instruction(i);
    if (!line.empty()) s += line + "\n";
  }
  return s;
}

// `v <= constant` / `v = v + 1` over the loop variable, the only loop shape the generator emits
// (/root/reference/src/generator.rs:401-456)
static bool is_var(const Expr& e, size_t id) { return e.kind == ExprKind::Var && e.var.id == id && !e.var.idx; }

std::string NoirTranspiler::transpile_instruction(const Instruction& instr) {
  switch (instr.kind) {
    case InstrKind::Assign: {
      const Variable& var = circuit.get_variable(instr.target);
      if (!(var.role == VariableRole::Local()) || var.var_type.is_array)
        throw std::logic_error("Assign target must be a local field");
      if (loop_vars_.count(var.var_ref.id)) throw NoirUnsupported("a for-loop variable is written in the loop body");
      return get_indent() + var.name + " = " + transpile_expr(*instr.value) + ";";
    }
    case InstrKind::ArrayAssign: {
      const Variable& var = circuit.get_variable(instr.target);
      if (!(var.role == VariableRole::Local()) || !var.var_type.is_array)
        throw std::logic_error("ArrayAssign target must be a local array");
      if (instr.index->kind != ExprKind::Constant) throw NoirUnsupported("array index is not a literal");
      return get_indent() + var.name + "[" + instr.index->constant.str() + "] = " + transpile_expr(*instr.value) + ";";
    }
    case InstrKind::Constraint: {
      const Variable& var = circuit.get_variable(instr.target);
      if (!var.role.is_output()) throw std::logic_error("Constraint target must be an output signal");
      const std::string idx = instr.target.idx ? "[" + std::to_string(*instr.target.idx) + "]" : "";
      return get_indent() + var.name + idx + " = " + transpile_expr(*instr.value) + ";";
    }
    case InstrKind::VarDecl: {
      const Variable& var = instr.var;
      if (var.role.is_signal) return "";  // inputs are parameters of main, outputs are declared at its top
      std::string init = var.var_type.is_array ? "[0; " + std::to_string(var.var_type.len) + "]" : "0";
      if (instr.default_value) init = transpile_expr(*instr.default_value);
      return get_indent() + "let mut " + var.name + ": " + type_of(var) + " = " + init + ";";
    }
    case InstrKind::If: {
      std::string res = get_indent() + "if " + transpile_expr(*instr.cond) + " != 0 {\n";
      indent_size_ += 1;
      res += transpile_block(instr.then_branch());
      indent_size_ -= 1;
      if (instr.else_branch()) {
        res += get_indent() + "} else {\n";
        indent_size_ += 1;
        res += transpile_block(*instr.else_branch());
        indent_size_ -= 1;
      }
      return res + get_indent() + "}";
    }
    case InstrKind::For: {
      if (!instr.init() || !instr.step() || !instr.cond || instr.init()->kind != InstrKind::VarDecl || !instr.init()->default_value ||
          instr.init()->default_value->kind != ExprKind::Constant)
        throw NoirUnsupported("for-loop does not start from a declared constant");
      const Variable& lv = instr.init()->var;
      const size_t id = lv.var_ref.id;
      const Expr& c = *instr.cond;
      if (c.kind != ExprKind::LessThanEq || !is_var(*c.left, id) || c.right->kind != ExprKind::Constant)
        throw NoirUnsupported("for-loop condition is not `variable <= constant`");
      const Instruction& st = *instr.step();
      if (st.kind != InstrKind::Assign || st.target.id != id || st.value->kind != ExprKind::Add || !is_var(*st.value->left, id) ||
          st.value->right->kind != ExprKind::Constant || st.value->right->constant.str() != "1")
        throw NoirUnsupported("for-loop step is not `variable = variable + 1`");
      if (instr.body().empty()) throw std::logic_error("Empty for loop bodies are not allowed");
      // Noir ranges exclude their end and count in an integer type: `lo..(hi + 1)`, with the
      // Field-typed loop variable of the circuit rebuilt from the counter in every trip
      std::string res = get_indent() + "for " + lv.name + "_i in " + instr.init()->default_value->constant.str() + "..(" +
                        c.right->constant.str() + " + 1) {\n";
      indent_size_ += 1;
      res += get_indent() + "let " + lv.name + ": Field = " + lv.name + "_i as Field;\n";
      loop_vars_.insert(id);
      res += transpile_block(instr.body());
      loop_vars_.erase(id);
      indent_size_ -= 1;
      return res + get_indent() + "}";
    }
    case InstrKind::While:
      throw NoirUnsupported("while loops do not exist in constrained Noir");
  }
  return "";
}

std::string NoirTranspiler::transpile_circuit() {
  std::string params, outs_decl, ret_type, ret_expr;
  size_t n_out = 0;
  for (const Variable& v : circuit.variables) {
    if (v.role.is_input()) {
      if (!params.empty()) params += ", ";
      params += v.name + ": " + type_of(v);
    } else if (v.role.is_output()) {
      outs_decl += "    let mut " + v.name + ": " + type_of(v) + " = " +
                   (v.var_type.is_array ? "[0; " + std::to_string(v.var_type.len) + "]" : std::string("0")) + ";\n";
      if (n_out++) {
        ret_type += ", ";
        ret_expr += ", ";
      }
      ret_type += type_of(v);
      ret_expr += v.name;
    }
  }
  std::string body = transpile_block(circuit.instructions);
  std::string s = "fn main(" + params + ")";
  if (n_out == 1) s += " -> pub " + ret_type;
  if (n_out > 1) s += " -> pub (" + ret_type + ")";
  s += " {\n" + outs_decl + body;
  if (n_out == 1) s += "    " + ret_expr + "\n";
  if (n_out > 1) s += "    (" + ret_expr + ")\n";
  return s + "}\n";
}

}  // namespace zkir
