#include "sexpr.hpp"

#include <cctype>
#include <map>
#include <stdexcept>

namespace zkir {
namespace {

const std::pair<ExprKind, const char*> kOps[] = {
    {ExprKind::Add, "add"}, {ExprKind::Sub, "sub"}, {ExprKind::Mul, "mul"}, {ExprKind::Power, "pow"},
    {ExprKind::Div, "div"}, {ExprKind::IntDiv, "idiv"}, {ExprKind::Rem, "rem"}, {ExprKind::LessThan, "lt"},
    {ExprKind::LessThanEq, "le"}, {ExprKind::GreaterThan, "gt"}, {ExprKind::GreaterThanEq, "ge"},
    {ExprKind::Equal, "eq"}, {ExprKind::And, "and"}, {ExprKind::Or, "or"}, {ExprKind::BitAnd, "band"},
    {ExprKind::BitOr, "bor"}, {ExprKind::BitXor, "bxor"}, {ExprKind::BitRShift, "shr"},
    {ExprKind::BitLShift, "shl"}, {ExprKind::BitNot, "bnot"}};

const char* op_name(ExprKind k) {
  for (auto& p : kOps)
    if (p.first == k) return p.second;
  return "?";
}

std::string quote(const std::string& s) {
  std::string r = "\"";
  for (char c : s) {
    if (c == '"' || c == '\\') r.push_back('\\');
    r.push_back(c);
  }
  return r + "\"";
}

void put_ref(std::string& o, const VarRef& r) {
  if (r.idx)
    o += "(" + std::to_string(r.id) + " " + std::to_string(*r.idx) + ")";
  else
    o += std::to_string(r.id);
}

void put_expr(std::string& o, const Expr& e) {
  switch (e.kind) {
    case ExprKind::Constant: o += "(const " + quote(e.constant.str()) + ")"; return;
    case ExprKind::Var: o += "(var "; put_ref(o, e.var); o += ")"; return;
    default: break;
  }
  o += "(";
  o += op_name(e.kind);
  o += " ";
  put_expr(o, *e.left);
  if (e.right) { o += " "; put_expr(o, *e.right); }
  o += ")";
}

void put_instr(std::string& o, const Instruction& i);
void put_block(std::string& o, const char* head, const std::vector<Instruction>& v) {
  o += "(";
  o += head;
  for (const Instruction& i : v) { o += " "; put_instr(o, i); }
  o += ")";
}

void put_instr(std::string& o, const Instruction& i) {
  switch (i.kind) {
    case InstrKind::VarDecl:
      o += "(decl " + std::to_string(i.var.var_ref.id);
      if (i.default_value) { o += " "; put_expr(o, *i.default_value); }
      o += ")";
      return;
    case InstrKind::Assign:
      o += "(assign "; put_ref(o, i.target); o += " "; put_expr(o, *i.value); o += ")";
      return;
    case InstrKind::ArrayAssign:
      o += "(aassign "; put_ref(o, i.target); o += " "; put_expr(o, *i.index); o += " "; put_expr(o, *i.value);
      o += ")";
      return;
    case InstrKind::Constraint:
      o += "(constrain "; put_ref(o, i.target); o += " "; put_expr(o, *i.value); o += ")";
      return;
    case InstrKind::If:
      o += "(if "; put_expr(o, *i.cond); o += " "; put_block(o, "then", i.then_branch());
      if (i.else_branch()) { o += " "; put_block(o, "else", *i.else_branch()); }
      o += ")";
      return;
    case InstrKind::For:
      o += "(for "; put_instr(o, *i.init()); o += " "; put_expr(o, *i.cond); o += " "; put_instr(o, *i.step());
      o += " "; put_block(o, "body", i.body()); o += ")";
      return;
    case InstrKind::While:
      o += "(while "; put_expr(o, *i.cond); o += " "; put_block(o, "body", i.body()); o += ")";
      return;
  }
}

// ---- reader ----
struct Node {
  bool is_list = false, is_string = false;
  std::string atom;
  std::vector<Node> items;
};

struct Reader {
  const std::string& s;
  size_t p = 0;
  explicit Reader(const std::string& t) : s(t) {}
  [[noreturn]] void fail(const std::string& m) const {
    throw std::runtime_error("circuit text, offset " + std::to_string(p) + ": " + m);
  }
  void ws() {
    while (p < s.size() && std::isspace((unsigned char)s[p])) p++;
  }
  Node read(int depth = 0) {
    if (depth > 4096) fail("nesting too deep");
    ws();
    if (p >= s.size()) fail("unexpected end");
    Node n;
    if (s[p] == '(') {
      p++;
      n.is_list = true;
      for (;;) {
        ws();
        if (p >= s.size()) fail("unclosed list");
        if (s[p] == ')') { p++; break; }
        n.items.push_back(read(depth + 1));
      }
    } else if (s[p] == '"') {
      p++;
      n.is_string = true;
      while (p < s.size() && s[p] != '"') {
        if (s[p] == '\\' && p + 1 < s.size()) p++;
        n.atom.push_back(s[p++]);
      }
      if (p >= s.size()) fail("unclosed string");
      p++;
    } else if (s[p] == ')') {
      fail("unexpected )");
    } else {
      while (p < s.size() && !std::isspace((unsigned char)s[p]) && s[p] != '(' && s[p] != ')') n.atom.push_back(s[p++]);
    }
    return n;
  }
};

[[noreturn]] void bad(const std::string& m) { throw std::runtime_error("circuit text: " + m); }

size_t to_index(const Node& n) {
  if (n.is_list || n.is_string || n.atom.empty() || n.atom.size() > 18) bad("expected a number");
  size_t v = 0;
  for (char c : n.atom) {
    if (c < '0' || c > '9') bad("expected a number, got " + n.atom);
    v = v * 10 + (size_t)(c - '0');
  }
  return v;
}
const std::string& head(const Node& n) {
  if (!n.is_list || n.items.empty() || n.items[0].is_list || n.items[0].is_string) bad("expected (name ...)");
  return n.items[0].atom;
}
VarRef to_ref(const Node& n) {
  VarRef r;
  if (n.is_list) {
    if (n.items.size() != 2) bad("reference must be id or (id idx)");
    r.id = to_index(n.items[0]);
    r.idx = to_index(n.items[1]);
  } else {
    r.id = to_index(n);
  }
  return r;
}

Expr to_expr(Arena& a, const Node& n) {
  const std::string& h = head(n);
  if (h == "const") {
    if (n.items.size() != 2 || !n.items[1].is_string) bad("(const \"decimal\")");
    return Expr::Constant(Field::from_text(a, n.items[1].atom));
  }
  if (h == "var") {
    if (n.items.size() != 2) bad("(var ref)");
    return Expr::Var(to_ref(n.items[1]));
  }
  for (auto& p : kOps)
    if (h == p.second) {
      if (p.first == ExprKind::BitNot) {
        if (n.items.size() != 2) bad("(bnot expr)");
        return Expr::Unary(a, p.first, to_expr(a, n.items[1]));
      }
      if (n.items.size() != 3) bad("(" + h + " expr expr)");
      return Expr::Binary(a, p.first, to_expr(a, n.items[1]), to_expr(a, n.items[2]));
    }
  bad("unknown expression " + h);
}

struct Parser {
  std::map<size_t, Variable> by_id;
  Arena* arena = nullptr;
  Instruction instr(const Node& n);
  std::vector<Instruction> block(const Node& n, const char* name) {
    if (head(n) != name) bad(std::string("expected (") + name + " ...)");
    std::vector<Instruction> v;
    for (size_t i = 1; i < n.items.size(); i++) v.push_back(instr(n.items[i]));
    return v;
  }
};

Instruction Parser::instr(const Node& n) {
  const std::string& h = head(n);
  const size_t k = n.items.size();
  if (h == "decl") {
    if (k != 2 && k != 3) bad("(decl id [expr])");
    auto it = by_id.find(to_index(n.items[1]));
    if (it == by_id.end()) bad("decl of an unknown variable id");
    std::optional<Expr> def;
    if (k == 3) def = to_expr(*arena, n.items[2]);
    return Instruction::VarDecl(*arena, it->second, def);
  }
  if (h == "assign") {
    if (k != 3) bad("(assign ref expr)");
    return Instruction::Assign(*arena, to_ref(n.items[1]), to_expr(*arena, n.items[2]));
  }
  if (h == "aassign") {
    if (k != 4) bad("(aassign ref index expr)");
    return Instruction::ArrayAssign(*arena, to_ref(n.items[1]), to_expr(*arena, n.items[2]), to_expr(*arena, n.items[3]));
  }
  if (h == "constrain") {
    if (k != 3) bad("(constrain ref expr)");
    return Instruction::Constraint(*arena, to_ref(n.items[1]), to_expr(*arena, n.items[2]));
  }
  if (h == "if") {
    if (k != 3 && k != 4) bad("(if cond (then ...) [(else ...)])");
    std::optional<std::vector<Instruction>> el;
    if (k == 4) el = block(n.items[3], "else");
    return Instruction::If(*arena, to_expr(*arena, n.items[1]), block(n.items[2], "then"), std::move(el));
  }
  if (h == "for") {
    if (k != 5) bad("(for init cond step (body ...))");
    return Instruction::For(*arena, instr(n.items[1]), to_expr(*arena, n.items[2]), instr(n.items[3]), block(n.items[4], "body"));
  }
  if (h == "while") {
    if (k != 3) bad("(while cond (body ...))");
    return Instruction::While(*arena, to_expr(*arena, n.items[1]), block(n.items[2], "body"));
  }
  bad("unknown instruction " + h);
}

}  // namespace

std::string print_circuit(const Circuit& c) {
  std::string o = "(circuit\n (vars";
  for (const Variable& v : c.variables) {
    o += "\n  (var " + std::to_string(v.var_ref.id) + " " + quote(v.name) + " ";
    o += v.var_type.is_array ? "(array " + std::to_string(v.var_type.len) + ")" : std::string("field");
    o += v.role.is_local() ? " local" : (v.role.is_input() ? " input" : " output");
    o += " (init";
    for (bool b : v.initialized) o += b ? " 1" : " 0";
    o += "))";
  }
  o += ")\n (instrs";
  for (const Instruction& i : c.instructions) { o += "\n  "; put_instr(o, i); }
  o += "))\n";
  return o;
}

Circuit parse_circuit(const std::string& text) {
  Reader rd(text);
  Node root = rd.read();
  rd.ws();
  if (rd.p != text.size()) rd.fail("trailing text");
  if (head(root) != "circuit" || root.items.size() != 3) bad("expected (circuit (vars ...) (instrs ...))");
  const Node& vars = root.items[1];
  const Node& instrs = root.items[2];
  if (head(vars) != "vars" || head(instrs) != "instrs") bad("expected (vars ...) then (instrs ...)");
  Parser ps;
  Circuit c;
  ps.arena = c.arena.get();
  for (size_t i = 1; i < vars.items.size(); i++) {
    const Node& n = vars.items[i];
    if (head(n) != "var" || n.items.size() != 6) bad("(var id \"name\" type role (init ...))");
    Variable v;
    v.var_ref.id = to_index(n.items[1]);
    if (!n.items[2].is_string) bad("variable name must be a string");
    v.name = n.items[2].atom;
    const Node& t = n.items[3];
    if (t.is_list) {
      if (head(t) != "array" || t.items.size() != 2) bad("(array n)");
      v.var_type = VariableType::Array(to_index(t.items[1]));
      if (v.var_type.len > (1u << 20)) bad("array too long");
    } else if (t.atom == "field") {
      v.var_type = VariableType::Field();
    } else {
      bad("type must be field or (array n)");
    }
    const std::string& role = n.items[4].atom;
    if (n.items[4].is_list) bad("role must be local, input or output");
    if (role == "local") v.role = VariableRole::Local();
    else if (role == "input") v.role = VariableRole::Signal(SignalType::Input);
    else if (role == "output") v.role = VariableRole::Signal(SignalType::Output);
    else bad("role must be local, input or output");
    if (head(n.items[5]) != "init") bad("(init ...)");
    for (size_t k = 1; k < n.items[5].items.size(); k++) v.initialized.push_back(to_index(n.items[5].items[k]) != 0);
    if (ps.by_id.count(v.var_ref.id)) bad("duplicate variable id");
    ps.by_id[v.var_ref.id] = v;
    c.variables.push_back(std::move(v));
  }
  for (size_t i = 1; i < instrs.items.size(); i++) c.instructions.push_back(ps.instr(instrs.items[i]));
  return c;
}

}  // namespace zkir
