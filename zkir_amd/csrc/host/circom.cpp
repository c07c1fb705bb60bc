// See circom.hpp.  Output is the reference's src/circom.rs output character for
// character: every expression fully parenthesised, three spaces per indent level,
// the `template Program()` wrapper with its leading newline and trailing spaces.
#include "circom.hpp"

#include <stdexcept>

namespace zkir {

std::string CircomTranspiler::get_indent() const {
  std::string s;
  for (size_t i = 0; i < indent_size_; i++) s += "   ";
  return s;
}

static const char* token_of(ExprKind k) {
  switch (k) {
    case ExprKind::Add: return "+";             // circom.rs:35-41
    case ExprKind::Sub: return "-";             // :42-46
    case ExprKind::Mul: return "*";             // :47-51
    case ExprKind::Power: return "**";          // :52-56
    case ExprKind::Div: return "/";             // :57-61
    case ExprKind::IntDiv: return "\\";         // :62-66
    case ExprKind::Rem: return "%";             // :67-71
    case ExprKind::LessThan: return "<";        // :72-78
    case ExprKind::LessThanEq: return "<=";     // :79-85
    case ExprKind::GreaterThan: return ">";     // :86-92
    case ExprKind::GreaterThanEq: return ">=";  // :93-99
    case ExprKind::Equal: return "==";          // :100-106
    case ExprKind::And: return "&&";            // :107-113
    case ExprKind::Or: return "||";             // :114-120
    case ExprKind::BitAnd: return "&";          // :122-128
    case ExprKind::BitOr: return "|";           // :129-135
    case ExprKind::BitXor: return "^";          // :137-143
    case ExprKind::BitRShift: return ">>";      // :144-150
    case ExprKind::BitLShift: return "<<";      // :151-157
    default: return "?";
  }
}

std::string CircomTranspiler::transpile_expr(const Expr& e) const {
  switch (e.kind) {
    case ExprKind::Var: {
      const Variable& var = circuit.get_variable(e.var);
      if (e.var.idx) return "(" + var.name + "[" + std::to_string(*e.var.idx) + "])";  // circom.rs:26-28
      return var.name;
    }
    case ExprKind::Constant:
      return e.constant.str();
    case ExprKind::BitNot:
      return "(~" + transpile_expr(*e.left) + ")";  // circom.rs:136
    default:
      return "(" + transpile_expr(*e.left) + token_of(e.kind) + transpile_expr(*e.right) + ")";
  }
}

std::string CircomTranspiler::transpile_block(const std::vector<Instruction>& body) {
  std::string s;
  for (size_t i = 0; i < body.size(); i++) {
    if (i) s += "\n";
    s += transpile_instruction(body[i]);
  }
  return s;
}

static std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && (s[a] == ' ' || s[a] == '\n' || s[a] == '\t' || s[a] == '\r')) a++;
  while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\n' || s[b - 1] == '\t' || s[b - 1] == '\r')) b--;
  return s.substr(a, b - a);
}

std::string CircomTranspiler::transpile_instruction(const Instruction& instr) {
  switch (instr.kind) {
    case InstrKind::Assign: {
      const Variable& var = circuit.get_variable(instr.target);
      if (!(var.role == VariableRole::Local())) throw std::logic_error("Assignment are only possible with variables");
      if (!(var.var_type == VariableType::Field())) throw std::logic_error("Assignment are only possible with fields");
      return get_indent() + var.name + "=" + transpile_expr(*instr.value) + ";";
    }
    case InstrKind::ArrayAssign: {
      const Variable& var = circuit.get_variable(instr.target);
      if (!(var.role == VariableRole::Local())) throw std::logic_error("Assignment is only possible between variables");
      if (!var.var_type.is_array) throw std::logic_error("ArrayAssign target is not an array");
      return get_indent() + var.name + "[" + transpile_expr(*instr.index) + "]=" + transpile_expr(*instr.value) + ";";
    }
    case InstrKind::Constraint: {
      const Variable& var = circuit.get_variable(instr.target);
      if (!var.role.is_output())
        throw std::logic_error("Constraint are only possible when the target is an output signal");
      return get_indent() + var.name + "<==" + transpile_expr(*instr.value) + ";";
    }
    case InstrKind::VarDecl: {
      const Variable& var = instr.var;
      std::string prefix = "var";
      if (var.role.is_signal) prefix = var.role.signal == SignalType::Input ? "signal input" : "signal output";
      std::string s = get_indent() + prefix + " " + var.name;
      if (var.var_type.is_array) s += "[" + std::to_string(var.var_type.len) + "]";
      if (instr.default_value) s += " = " + transpile_expr(*instr.default_value);
      return s + ";";
    }
    case InstrKind::If: {
      std::string res = get_indent() + "if (" + transpile_expr(*instr.cond) + ") {\n";
      indent_size_ += 1;
      res += transpile_block(instr.then_branch()) + "\n";
      indent_size_ -= 1;
      if (instr.else_branch()) {
        res += get_indent() + "} else {\n";
        indent_size_ += 1;
        res += transpile_block(*instr.else_branch()) + "\n";
        indent_size_ -= 1;
      }
      res += get_indent() + "}";
      return res;
    }
    case InstrKind::For: {
      if (!instr.init() || !instr.step()) throw std::logic_error("For-loop without init or step");
      if (!(instr.init()->kind == InstrKind::Assign || instr.init()->kind == InstrKind::VarDecl))
        throw std::logic_error("For-loop inits must be an assignment");
      if (instr.step()->kind != InstrKind::Assign) throw std::logic_error("For-loop steps must be an assignment");
      if (instr.body().empty()) throw std::logic_error("Empty for loop bodies are not allowed");
      // init keeps its `;`, step loses every `;` (circom.rs:354-362)
      std::string init = trim(transpile_instruction(*instr.init()));
      std::string cond = transpile_expr(*instr.cond);
      std::string step_raw = transpile_instruction(*instr.step()), step;
      for (char ch : step_raw)
        if (ch != ';') step.push_back(ch);
      step = trim(step);
      std::string res = get_indent() + "for (" + init + " " + cond + "; " + step + ") {\n";
      indent_size_ += 1;
      res += transpile_block(instr.body()) + "\n";
      indent_size_ -= 1;
      res += get_indent() + "}";
      return res;
    }
    case InstrKind::While: {
      if (instr.body().empty()) throw std::logic_error("Empty while loop bodies are not allowed");
      std::string res = get_indent() + "while (" + transpile_expr(*instr.cond) + ") {\n";
      indent_size_ += 1;
      res += transpile_block(instr.body()) + "\n";
      indent_size_ -= 1;
      res += get_indent() + "}";
      return res;
    }
  }
  return "";
}

std::string CircomTranspiler::transpile_circuit() {
  std::string instructions;
  for (const Instruction& ins : circuit.instructions) instructions += transpile_instruction(ins) + "\n";
  // the raw string of circom.rs:421-428: leading newline, trailing eight spaces
  return "\ntemplate Program() {\n" + instructions + "}\n\ncomponent main = Program();\n        ";
}

// ---- `{:#?}` ---------------------------------------------------------------
// Rust's pretty Debug: one field or element per line, four spaces per level, a
// trailing comma after every item, tuple variants as `Name(` ... `)`, struct
// variants as `Name {` ... `}`, Option as `Some(` ... `)` / `None`, strings quoted.
namespace {

struct Dbg {
  std::string out;
  int level = 0;
  void pad() { out.append((size_t)level * 4, ' '); }
  void open(const std::string& head) { out += head + "\n"; level++; }
  void close(const std::string& tail) { level--; pad(); out += tail; }
  void item_end() { out += ",\n"; }
};

std::string quote(const std::string& s) {
  std::string r = "\"";
  for (char c : s) {
    if (c == '"' || c == '\\') r.push_back('\\');
    r.push_back(c);
  }
  return r + "\"";
}

void dbg_varref(Dbg& d, const VarRef& v) {
  d.open("VarRef {");
  d.pad(); d.out += "id: " + std::to_string(v.id); d.item_end();
  d.pad();
  if (v.idx) {
    d.out += "idx: ";
    d.open("Some(");
    d.pad(); d.out += std::to_string(*v.idx); d.item_end();
    d.close(")");
  } else {
    d.out += "idx: None";
  }
  d.item_end();
  d.close("}");
}

const char* expr_name(ExprKind k) {
  switch (k) {
    case ExprKind::Var: return "Var";
    case ExprKind::Constant: return "Constant";
    case ExprKind::Add: return "Add";
    case ExprKind::Sub: return "Sub";
    case ExprKind::Mul: return "Mul";
    case ExprKind::Power: return "Power";
    case ExprKind::Div: return "Div";
    case ExprKind::IntDiv: return "IntDiv";
    case ExprKind::Rem: return "Rem";
    case ExprKind::LessThan: return "LessThan";
    case ExprKind::LessThanEq: return "LessThanEq";
    case ExprKind::GreaterThan: return "GreaterThan";
    case ExprKind::GreaterThanEq: return "GreaterThanEq";
    case ExprKind::Equal: return "Equal";
    case ExprKind::And: return "And";
    case ExprKind::Or: return "Or";
    case ExprKind::BitAnd: return "BitAnd";
    case ExprKind::BitOr: return "BitOr";
    case ExprKind::BitNot: return "BitNot";
    case ExprKind::BitXor: return "BitXor";
    case ExprKind::BitRShift: return "BitRShift";
    case ExprKind::BitLShift: return "BitLShift";
  }
  return "?";
}

void dbg_expr(Dbg& d, const Expr& e) {
  d.open(std::string(expr_name(e.kind)) + "(");
  if (e.kind == ExprKind::Var) {
    d.pad(); dbg_varref(d, e.var); d.item_end();
  } else if (e.kind == ExprKind::Constant) {
    d.pad(); d.out += quote(e.constant.str()); d.item_end();
  } else {
    d.pad(); dbg_expr(d, *e.left); d.item_end();
    if (e.right) { d.pad(); dbg_expr(d, *e.right); d.item_end(); }
  }
  d.close(")");
}

void dbg_variable(Dbg& d, const Variable& v) {
  d.open("Variable {");
  d.pad(); d.out += "var_ref: "; dbg_varref(d, v.var_ref); d.item_end();
  d.pad(); d.out += "name: " + quote(v.name); d.item_end();
  d.pad(); d.out += "var_type: ";
  if (v.var_type.is_array) {
    d.open("Array(");
    d.pad(); d.out += std::to_string(v.var_type.len); d.item_end();
    d.close(")");
  } else {
    d.out += "Field";
  }
  d.item_end();
  d.pad(); d.out += "role: ";
  if (v.role.is_signal) {
    d.open("Signal(");
    d.pad(); d.out += v.role.signal == SignalType::Input ? "Input" : "Output"; d.item_end();
    d.close(")");
  } else {
    d.out += "Local";
  }
  d.item_end();
  d.pad(); d.out += "initialized: ";
  d.open("[");
  for (bool b : v.initialized) { d.pad(); d.out += b ? "true" : "false"; d.item_end(); }
  d.close("]");
  d.item_end();
  d.close("}");
}

void dbg_instr(Dbg& d, const Instruction& i);

void dbg_vec(Dbg& d, const std::vector<Instruction>& v) {
  if (v.empty()) { d.out += "[]"; return; }
  d.open("[");
  for (const Instruction& i : v) { d.pad(); dbg_instr(d, i); d.item_end(); }
  d.close("]");
}

void dbg_opt_expr(Dbg& d, const Expr* e) {
  if (!e) { d.out += "None"; return; }
  d.open("Some(");
  d.pad(); dbg_expr(d, *e); d.item_end();
  d.close(")");
}

void dbg_instr(Dbg& d, const Instruction& i) {
  switch (i.kind) {
    case InstrKind::VarDecl:
      d.open("VarDecl(");
      d.pad(); dbg_variable(d, i.var); d.item_end();
      d.pad(); dbg_opt_expr(d, i.default_value); d.item_end();
      d.close(")");
      break;
    case InstrKind::Assign:
      d.open("Assign(");
      d.pad(); dbg_varref(d, i.target); d.item_end();
      d.pad(); dbg_expr(d, *i.value); d.item_end();
      d.close(")");
      break;
    case InstrKind::ArrayAssign:
      d.open("ArrayAssign(");
      d.pad(); dbg_varref(d, i.target); d.item_end();
      d.pad(); dbg_expr(d, *i.index); d.item_end();
      d.pad(); dbg_expr(d, *i.value); d.item_end();
      d.close(")");
      break;
    case InstrKind::Constraint:
      d.open("Constraint(");
      d.pad(); dbg_varref(d, i.target); d.item_end();
      d.pad(); dbg_expr(d, *i.value); d.item_end();
      d.close(")");
      break;
    case InstrKind::If:
      d.open("If {");
      d.pad(); d.out += "cond: "; dbg_expr(d, *i.cond); d.item_end();
      d.pad(); d.out += "then_branch: "; dbg_vec(d, i.then_branch()); d.item_end();
      d.pad(); d.out += "else_branch: ";
      if (i.else_branch()) {
        d.open("Some(");
        d.pad(); dbg_vec(d, *i.else_branch()); d.item_end();
        d.close(")");
      } else {
        d.out += "None";
      }
      d.item_end();
      d.close("}");
      break;
    case InstrKind::For:
      d.open("For {");
      d.pad(); d.out += "init: "; dbg_instr(d, *i.init()); d.item_end();
      d.pad(); d.out += "cond: "; dbg_expr(d, *i.cond); d.item_end();
      d.pad(); d.out += "step: "; dbg_instr(d, *i.step()); d.item_end();
      d.pad(); d.out += "body: "; dbg_vec(d, i.body()); d.item_end();
      d.close("}");
      break;
    case InstrKind::While:
      d.open("While {");
      d.pad(); d.out += "cond: "; dbg_expr(d, *i.cond); d.item_end();
      d.pad(); d.out += "body: "; dbg_vec(d, i.body()); d.item_end();
      d.close("}");
      break;
  }
}

}  // namespace

std::string debug_instructions(const std::vector<Instruction>& instrs) {
  Dbg d;
  dbg_vec(d, instrs);
  return d.out;
}

}  // namespace zkir
