// See generator.hpp.  Function by function this follows the reference's
// src/generator.rs; the order of every rng_ call is the reference's order, because
// a seed must reproduce the same circuit here and there.
#include "generator.hpp"

#include <algorithm>

#include <string>

namespace zkir {

GeneratorConfig default_generator_config() {
  GeneratorConfig c{};
  c.min_instructions = 15;  // generator.rs:147
  c.max_instructions = 30;
  c.statement_budget = 10;  // generator.rs:167
  c.max_array_len = 10;     // generator.rs:224
  c.max_expr_depth = 1;     // generator.rs:486
  c.loop_init_max = 30;     // generator.rs:406
  c.loop_bound_max = 60;    // generator.rs:416
  c.loop_body_max = 5;      // generator.rs:436
  c.name_len = 8;           // generator.rs:264
  return c;
}

void validate_generator_config(const GeneratorConfig& c) {
  if (c.min_instructions > c.max_instructions || c.max_array_len == 0 || c.loop_body_max == 0 ||
      c.max_instructions > (1u << 22) || c.name_len == 0 || c.name_len > 64 || c.max_expr_depth > 64 ||
      c.loop_init_max > c.loop_bound_max)
    throw std::invalid_argument("generator config out of range");
}

// ---- ScopeStack (generator.rs:48-115) ----
void Generator::ScopeStack::refresh(size_t pos) {
  const Variable& v = flat[pos];
  const bool field = !v.var_type.is_array;
  ready_fields.set(pos, field && v.all_initialized());
  open_outputs.set(pos, v.role.is_output() && field && !v.all_initialized());
  const uint32_t written = (uint32_t)v.initialized.count();
  written_count[pos] = written;
  const bool touched = !field && written > 0;
  if (touched && !touched_arrays.test(pos))  // first write to this array: keep the list ascending
    touched_list.insert(std::upper_bound(touched_list.begin(), touched_list.end(), pos), pos);
  touched_arrays.set(pos, touched);
  ready_signals.set(pos, v.role.is_signal && field && v.all_initialized());
}

Variable Generator::ScopeStack::add_var(std::string name, VariableType t, VariableRole r,
                                        InitFlags initialized) {
  Variable var;
  var.var_ref.id = next_id;
  var.name = std::move(name);
  var.var_type = t;
  var.role = r;
  var.initialized = std::move(initialized);
  const size_t pos = flat.size();
  flat.push_back(var);
  all_vars.push_back(var);  // a copy taken now: later mark_initialized calls do not reach it (generator.rs:84)
  next_id += 1;
  if (r.is_local()) {
    n_local++;
    (t.is_array ? local_arrays : local_fields).push_back(pos);
  }
  if (r.is_output()) n_output++;
  for (OrderedSet* s : {&ready_fields, &open_outputs, &touched_arrays, &ready_signals}) s->resize(pos + 1);
  written_count.resize(pos + 1, 0);
  refresh(pos);
  return var;
}

void Generator::ScopeStack::leave_scope() {
  const size_t begin = scope_begin.back();
  scope_begin.pop_back();
  for (size_t p = begin; p < flat.size(); p++) {
    if (flat[p].role.is_local()) n_local--;
    if (flat[p].role.is_output()) n_output--;
  }
  while (!local_fields.empty() && local_fields.back() >= begin) local_fields.pop_back();
  while (!local_arrays.empty() && local_arrays.back() >= begin) local_arrays.pop_back();
  for (OrderedSet* s : {&ready_fields, &open_outputs, &touched_arrays, &ready_signals}) s->resize(begin);
  while (!touched_list.empty() && touched_list.back() >= begin) touched_list.pop_back();
  written_count.resize(begin);
  flat.resize(begin);
}

void Generator::ScopeStack::mark_initialized(const VarRef& var_id) {
  // ids grow with the position in `flat`: binary search
  size_t lo = 0, hi = flat.size();
  while (lo < hi) {
    const size_t mid = (lo + hi) / 2;
    if (flat[mid].var_ref.id < var_id.id) lo = mid + 1; else hi = mid;
  }
  if (lo == flat.size() || flat[lo].var_ref.id != var_id.id) return;  // not in scope: nothing to mark (generator.rs:101-114)
  Variable& var = flat[lo];
  if (var_id.idx)
    var.initialized.set(*var_id.idx, true);
  else
    var.initialized.assign(1, true);
  refresh(lo);
}

// ---- Generator ----
void Generator::push_generated(std::vector<Instruction>& out, std::optional<Instruction>&& instr) {
  if (instr) {
    for (Instruction& p : pending_) out.push_back(std::move(p));
    out.push_back(std::move(*instr));
  }
  pending_.clear();
}

Circuit Generator::generate() {
  std::vector<Instruction> instructions;
  scope_stack_.enter_scope();

  const size_t n = rng_.rand(cfg_.min_instructions, cfg_.max_instructions);
  requested_ = n;
  instructions.reserve(n + n / 2 + cfg_.forced_inputs);
  // at most one declaration per statement, plus the output a quadratic-aware constraint declares for itself
  scope_stack_.flat.reserve(n + cfg_.forced_inputs + 16);
  scope_stack_.all_vars.reserve(n + cfg_.forced_inputs + 16);

  // extension: scalar input signals declared before anything else
  for (uint32_t k = 0; k < cfg_.forced_inputs; k++) {
    auto d = gen_var_decl(VariableType::Field(), VariableRole::Signal(SignalType::Input), std::nullopt);
    instructions.push_back(std::move(*d));
  }

  for (size_t i = 0; i < n; i++) {
    std::optional<Instruction> instr = generate_instr();
    const bool some = instr.has_value();
    push_generated(instructions, std::move(instr));
    if (some) n_instr_ = 0;
  }
  generated_ = instructions.size();

  scope_stack_.leave_scope();
  if (!scope_stack_.empty()) throw std::logic_error("scope stack not empty");  // generator.rs:161
  return Circuit(std::move(scope_stack_.all_vars), std::move(instructions), arena_);  // a Generator is single use
}

std::optional<Instruction> Generator::generate_instr() {
  if (n_instr_ >= cfg_.statement_budget) return std::nullopt;
  n_instr_ += 1;

  const bool has_vars = scope_stack_.n_local > 0;                // any Local in scope (generator.rs:174-177)
  const bool has_arrays = !scope_stack_.local_arrays.empty();    // any Local array (:178-182)
  const bool has_output_signals = scope_stack_.n_output > 0;     // any Output signal (:183-186)

  // candidate order of generator.rs:188-204
  enum Cand { kVarDecl, kIf, kFor, kAssign, kArrayAssign, kConstraint };
  Cand funcs[6];
  size_t n_funcs = 0;
  funcs[n_funcs++] = kVarDecl;
  if (cfg_.enable_if) funcs[n_funcs++] = kIf;  // extension: the reference keeps this entry commented out (:190)
  if (!cfg_.no_loops) funcs[n_funcs++] = kFor;
  if (has_vars) funcs[n_funcs++] = kAssign;
  if (has_arrays) funcs[n_funcs++] = kArrayAssign;
  if (has_output_signals || cfg_.quadratic_aware) funcs[n_funcs++] = kConstraint;

  const size_t idx = rng_.rand(0, n_funcs - 1);
  switch (funcs[idx]) {
    case kVarDecl: return gen_var_decl_default();
    case kIf: return gen_if();
    case kFor: return gen_for();
    case kAssign: return gen_assign();
    case kArrayAssign: return gen_array_assign();
    case kConstraint: return cfg_.quadratic_aware ? gen_constraint_quadratic() : gen_constraint();
  }
  return std::nullopt;
}

std::optional<Instruction> Generator::gen_var_decl(std::optional<VariableType> type_kind,
                                                   std::optional<VariableRole> role_kind,
                                                   std::optional<Expr> default_value) {
  VariableType var_type;
  if (type_kind) {
    var_type = *type_kind;
  } else if (rng_.rand(0, 1) == 0) {
    var_type = VariableType::Field();
  } else {
    var_type = VariableType::Array(rng_.rand(1, cfg_.max_array_len));
  }

  // 9/10 chance of declaring a variable with a default value.  A caller that asks for
  // a signal (only the extensions do) gets none: signals cannot take a default.
  const bool wants_signal = role_kind && role_kind->is_signal;
  if (!default_value && !var_type.is_array && !wants_signal && rng_.rand(1, 10) != 1) {
    std::optional<Expr> val = generate_expr(0);
    if (!val) return std::nullopt;
    default_value = std::move(val);
  }

  VariableRole role;
  if (role_kind) {
    role = *role_kind;
  } else if (default_value) {
    role = VariableRole::Local();  // only locals can take a default value
  } else if (rng_.rand(0, 1) == 0) {
    role = rng_.rand(0, 1) == 0 ? VariableRole::Signal(SignalType::Input) : VariableRole::Signal(SignalType::Output);
    // extension: when every constraint declares its own output, a free-standing output
    // could never be constrained; make it a local instead
    if (cfg_.quadratic_aware && role.is_output()) role = VariableRole::Local();
    // extension: with a fixed set of up-front inputs, no further inputs appear
    if (cfg_.forced_inputs && role.is_input()) role = VariableRole::Local();
  } else {
    role = VariableRole::Local();
  }

  // all false even when a default is present (generator.rs:255-260); the extension
  // makes input signals readable from the start
  const bool readable = cfg_.inputs_readable && role.is_input();
  InitFlags initialized(var_type.is_array ? var_type.len : 1, readable);

  std::string name = rng_.rand_string(cfg_.name_len);
  Variable v = scope_stack_.add_var(std::move(name), var_type, role, std::move(initialized));
  return Instruction::VarDecl(*arena_, std::move(v), default_value);
}

std::optional<Instruction> Generator::gen_assign() {
  // rand_vec over the Local Field variables in scope order (generator.rs:270-281)
  const std::vector<size_t>& vars = scope_stack_.local_fields;
  const size_t* pick = rng_.rand_vec(vars);
  if (!pick) return std::nullopt;
  VarRef var_id = scope_stack_.flat[*pick].var_ref;
  std::optional<Expr> expr = generate_expr(0);
  if (!expr) return std::nullopt;

  scope_stack_.mark_initialized(var_id);
  return Instruction::Assign(*arena_, var_id, *expr);
}

std::optional<Instruction> Generator::gen_array_assign() {
  // rand_vec over the Local arrays in scope order (generator.rs:292-306)
  if (const size_t* pick = rng_.rand_vec(scope_stack_.local_arrays)) {
    const VarRef var_ref = scope_stack_.flat[*pick].var_ref;
    const size_t n = scope_stack_.flat[*pick].var_type.len;
    std::optional<Expr> expr = generate_expr(0);
    if (expr) {
      const size_t idx = rng_.rand(0, n - 1);
      VarRef marked;
      marked.id = var_ref.id;
      marked.idx = idx;
      scope_stack_.mark_initialized(marked);
      return Instruction::ArrayAssign(*arena_, var_ref, Expr::Constant((uint64_t)idx), *expr);
    }
  }
  return std::nullopt;
}

std::optional<Instruction> Generator::gen_constraint() {
  // rand_vec over the Output Field signals not yet initialised, in scope order (generator.rs:330-343)
  const size_t n_open = scope_stack_.open_outputs.count();
  if (n_open) {
    const VarRef var_id = scope_stack_.flat[scope_stack_.open_outputs.select(rng_.rand(0, n_open - 1))].var_ref;
    // marked before the right-hand side is generated (generator.rs:344-346): `out <== out` is reachable
    scope_stack_.mark_initialized(var_id);
    std::optional<Expr> expr = generate_expr(0);
    if (expr) return Instruction::Constraint(*arena_, var_id, *expr);
  }
  return std::nullopt;
}

std::optional<Instruction> Generator::gen_if() {
  std::optional<Expr> cond;
  if (cfg_.enable_if == 2) {
    // extension: a comparison of two constants, decidable when the circuit is lowered
    static const ExprKind kCmp[] = {ExprKind::LessThan, ExprKind::Equal, ExprKind::GreaterThan};
    const ExprKind k = kCmp[rng_.rand(0, 2)];
    Expr l = Expr::Constant(rng_.next());
    Expr r = Expr::Constant(rng_.next());
    cond = Expr::Binary(*arena_, k, l, r);
  } else {
    cond = generate_expr(0);
  }
  if (!cond) return std::nullopt;

  scope_stack_.enter_scope();
  std::vector<Instruction> then_branch;
  for (size_t i = 0, n = rng_.rand(1, cfg_.loop_body_max); i < n; i++) push_generated(then_branch, generate_instr());
  scope_stack_.leave_scope();
  if (then_branch.empty()) return std::nullopt;

  scope_stack_.enter_scope();
  std::optional<std::vector<Instruction>> else_branch;
  if (rng_.rand(0, 1) == 0) {  // an else branch half of the time
    std::vector<Instruction> instrs;
    for (size_t i = 0, n = rng_.rand(1, cfg_.loop_body_max); i < n; i++) push_generated(instrs, generate_instr());
    if (!instrs.empty()) else_branch = std::move(instrs);
  }
  scope_stack_.leave_scope();
  return Instruction::If(*arena_, *cond, std::move(then_branch), std::move(else_branch));
}

std::optional<Instruction> Generator::gen_for() {
  scope_stack_.enter_scope();

  // the loop always declares a fresh variable with a constant start value
  const size_t default_value = rng_.rand(0, cfg_.loop_init_max);
  Instruction init = *gen_var_decl(VariableType::Field(), VariableRole::Local(),
                                   Expr::Constant((uint64_t)default_value));
  const size_t max = rng_.rand(default_value, cfg_.loop_bound_max);
  const VarRef var_ref = init.var.var_ref;

  Expr cond = Expr::Binary(*arena_, ExprKind::LessThanEq, Expr::Var(var_ref), Expr::Constant((uint64_t)max));
  Instruction step =
      Instruction::Assign(*arena_, var_ref, Expr::Binary(*arena_, ExprKind::Add, Expr::Var(var_ref), Expr::Constant((uint64_t)1)));

  std::vector<Instruction> body;
  for (size_t i = 0, n = rng_.rand(1, cfg_.loop_body_max); i < n; i++) push_generated(body, generate_instr());

  scope_stack_.leave_scope();
  if (body.empty()) return std::nullopt;  // the loop variable stays in all_vars, the draws are spent
  return Instruction::For(*arena_, std::move(init), cond, std::move(step), std::move(body));
}

std::optional<Instruction> Generator::gen_while() {
  std::optional<Expr> cond = generate_expr(0);
  if (!cond) return std::nullopt;
  scope_stack_.enter_scope();
  std::vector<Instruction> body;
  for (size_t i = 0, n = rng_.rand(1, cfg_.loop_body_max); i < n; i++) push_generated(body, generate_instr());
  scope_stack_.leave_scope();
  if (body.empty()) return std::nullopt;
  return Instruction::While(*arena_, *cond, std::move(body));
}

std::optional<Expr> Generator::gen_bin_expr(ExprKind k, size_t depth) {
  std::optional<Expr> left = generate_expr(depth + 1);
  std::optional<Expr> right = generate_expr(depth + 1);
  if (!left || !right) return std::nullopt;
  return Expr::Binary(*arena_, k, *left, *right);
}

std::optional<Expr> Generator::gen_un_expr(ExprKind k, size_t depth) {
  std::optional<Expr> e = generate_expr(depth + 1);
  if (!e) return std::nullopt;
  return Expr::Unary(*arena_, k, *e);
}

std::optional<Expr> Generator::generate_expr(size_t depth) {
  // hard limit (generator.rs:486-488)
  if (depth >= cfg_.max_expr_depth) return Expr::Constant(rng_.next());
  // 1/3 chance of stopping early; unreachable at the reference's limit of 1 (generator.rs:490-493)
  if (depth >= 1 && rng_.rand(1, 3) == 1) return Expr::Constant(rng_.next());

  const bool has_field_vars = scope_stack_.ready_fields.count() > 0;    // generator.rs:496-499
  const bool has_array_vars = scope_stack_.touched_arrays.count() > 0;  // generator.rs:502-505

  // the fixed entries of generator.rs:507-531, in order; -1 stands for gen_constant
  static const int kAll[] = {
      -1,
      (int)ExprKind::Add, (int)ExprKind::Sub, (int)ExprKind::Mul, (int)ExprKind::Power, (int)ExprKind::Div,
      (int)ExprKind::IntDiv, (int)ExprKind::Rem, (int)ExprKind::LessThan, (int)ExprKind::LessThanEq,
      (int)ExprKind::GreaterThan, (int)ExprKind::GreaterThanEq, (int)ExprKind::Equal, (int)ExprKind::And,
      (int)ExprKind::Or, (int)ExprKind::BitAnd, (int)ExprKind::BitOr, (int)ExprKind::BitNot, (int)ExprKind::BitXor,
      (int)ExprKind::BitRShift, (int)ExprKind::BitLShift};
  static const int kArith[] = {-1, (int)ExprKind::Add, (int)ExprKind::Sub, (int)ExprKind::Mul};
  const int* table = cfg_.arithmetic_only ? kArith : kAll;
  const size_t n_fixed = cfg_.arithmetic_only ? sizeof(kArith) / sizeof(int) : sizeof(kAll) / sizeof(int);

  const size_t n_funcs = n_fixed + (has_field_vars ? 1 : 0) + (has_array_vars ? 1 : 0);
  const size_t idx = rng_.rand(0, n_funcs - 1);
  if (idx < n_fixed) {
    const int f = table[idx];
    if (f < 0) return gen_constant(depth);
    if ((ExprKind)f == ExprKind::BitNot) return gen_un_expr(ExprKind::BitNot, depth);
    return gen_bin_expr((ExprKind)f, depth);
  }
  if (has_field_vars && idx == n_fixed) return gen_var_ref(depth);
  return gen_array_index(depth);
}

std::optional<Expr> Generator::gen_var_ref(size_t) {
  const size_t n = scope_stack_.ready_fields.count();
  if (!n) return std::nullopt;
  return Expr::Var(scope_stack_.flat[scope_stack_.ready_fields.select(rng_.rand(0, n - 1))].var_ref);
}

std::optional<Expr> Generator::gen_constant(size_t) { return Expr::Constant(rng_.next()); }

std::optional<Expr> Generator::gen_array_index(size_t) {
  // One draw per array that has a written element, in scope order, then one draw to choose
  // among them (generator.rs:572-601); only the chosen array's draw is used, but every draw
  // must be made to keep the random stream aligned with the reference.
  const std::vector<size_t>& arrays = scope_stack_.touched_list;
  const size_t n = arrays.size();
  if (n == 0) return std::nullopt;
  // a position in 0..=written, not one of the written elements (generator.rs:584): it can
  // name an unwritten element, or one past the end when every element is written
  // rand(0, w) with w >= 1 always draws (an array is in the list once an element is written), and
  // is next() % (w + 1): the stream advances once per array, and only the chosen array's draw is looked at
  const Rng before = rng_;
  rng_.skip(n);
  const size_t pick = rng_.rand(0, n - 1);
  const size_t chosen_pos = arrays[pick];
  size_t chosen_idx = before.peek(pick) % ((uint64_t)scope_stack_.written_count[chosen_pos] + 1);
  const Variable& v = scope_stack_.flat[chosen_pos];
  if (cfg_.safe_array_index) {  // extension: always a written element
    const size_t written = v.initialized.count();
    size_t want = chosen_idx % written, seen = 0;
    for (size_t i = 0; i < v.initialized.size(); i++)
      if (v.initialized.get(i) && seen++ == want) {
        chosen_idx = i;
        break;
      }
  }
  VarRef r;
  r.id = v.var_ref.id;
  r.idx = chosen_idx;
  return Expr::Var(r);
}

// ---- extensions (no counterpart in the reference) ----

// a*x + b style term over readable signals and constants: degree <= 1 in the signals
Expr Generator::gen_linear(bool factor) {
  const OrderedSet& sigs = scope_stack_.ready_signals;
  const bool always_signal = factor && cfg_.signal_factors;  // extension: a factor never draws a constant
  auto term = [&]() -> Expr {
    if (sigs.count() && (always_signal || rng_.rand(0, 3) != 0)) {
      size_t k = rng_.rand(0, sigs.count() - 1);
      // extension: with a signal window, a draw that falls on an older signal is folded onto
      // the forced inputs and the most recent `signal_window` signals
      const size_t keep = (size_t)cfg_.forced_inputs + cfg_.signal_window;
      if (cfg_.signal_window && sigs.count() > keep && k >= cfg_.forced_inputs && k < sigs.count() - cfg_.signal_window)
        k = (k % keep < cfg_.forced_inputs) ? k % keep : sigs.count() - keep + k % keep;
      return Expr::Var(scope_stack_.flat[sigs.select(k)].var_ref);
    }
    return Expr::Constant(rng_.next());
  };
  Expr t = term();
  if (cfg_.max_linear_terms == 1) return t;
  const uint64_t shape = rng_.rand(0, 2);
  if (shape == 0) return t;
  return Expr::Binary(*arena_, shape == 1 ? ExprKind::Add : ExprKind::Sub, t, term());
}

// `fresh output <== lin * lin + lin`: a real rank-1 constraint
std::optional<Instruction> Generator::gen_constraint_quadratic() {
  Expr a = gen_linear(true), b = gen_linear(true), c = gen_linear(false);
  auto decl = gen_var_decl(VariableType::Field(), VariableRole::Signal(SignalType::Output), std::nullopt);
  const VarRef out = decl->var.var_ref;
  pending_.push_back(std::move(*decl));
  scope_stack_.mark_initialized(out);
  Expr rhs = Expr::Binary(*arena_, ExprKind::Add, Expr::Binary(*arena_, ExprKind::Mul, a, b), c);
  return Instruction::Constraint(*arena_, out, rhs);
}

}  // namespace zkir
