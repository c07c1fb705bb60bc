// See lower.hpp.  Two passes:
//   1. an abstract interpreter walks the Instruction tree, unrolling loops and
//      deciding every condition and index from values known at lowering time (the
//      way circom decides them at compile time), and emits single-assignment ops;
//   2. a linear scan gives every value a home: the accumulator when its uses
//      follow immediately, an on-chip (LDS) slot when one is free, else a
//      device-memory slot; slots are reused once a value is dead.
#include "lower.hpp"

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <queue>
#include <unordered_map>

#include "fr_host.hpp"
#include "generator.hpp"
#include "ordered_set.hpp"
#include "rng.hpp"

namespace zkir {

LowerOptions default_lower_options() {
  LowerOptions o{};
  o.fold_constants = 1;
  o.n_lslots = 2;  // with the victim row of the generated-stream kernel: three rows of LDS, and five workgroups fit a CU
  o.reuse_slots = 1;
  o.max_ops = 0;
  o.bnot_variant = 0;
  o.use_acc = 1;
  o.reorder = 1;
  o.window_cache = 1;
  o.order_ops = 1;
  o.scaled_values = 1;
  return o;
}

namespace {

// operand of a single-assignment op: >= 0 a value id, < 0 constant ~index
typedef int32_t Ref;
constexpr Ref kNoRef = INT32_MIN;
inline Ref const_ref(uint32_t idx) { return ~(Ref)idx; }
inline bool is_const(Ref r) { return r < 0 && r != kNoRef; }
inline uint32_t const_index(Ref r) { return (uint32_t)~r; }

struct SOp {
  uint8_t code;
  Ref dst;   // value defined (value ops); output index (OUTPUT); `c` (CHECK)
  Ref a, b;  // INPUT: a = input element index
};

enum Degree { kKnown = 0, kLinear = 1, kQuadratic = 2, kNonQuadratic = 3 };

// one term of the linear part of a quadratic value: coef * ref
struct Term {
  U256 coef;
  Ref ref;
};

struct Val {
  bool assigned = false;  // false: an output signal nothing has been assigned to yet
  bool known = false;     // value is fixed at lowering time
  bool from_signal = false;  // derives from a signal: circom does not know it at compile time even where this
                             // pass does (a signal constrained to a constant), so it may not decide a condition or an index
  U256 value;             // when known
  Ref ref = kNoRef;       // where the tape holds it (may be absent for a folded known value)
  Degree deg = kKnown;
  int32_t quad = -1;      // deg == kQuadratic: which entry of Pass1::quads_ describes it
};

// a quadratic value: scale * qa * qb + sum(terms).  Kept beside the values (one per product of two linear values and
// per sum built on it), so that a Val is plain data: expression evaluation copies one per node.
struct Quad {
  Ref qa = kNoRef, qb = kNoRef;
  U256 scale;
  std::vector<Term> terms;
};

struct Cells {
  size_t id = 0;
  VariableRole role;
  bool is_array = false;
  size_t type_len = 0;                // Array(type_len)
  const std::string* name = nullptr;  // in circuit.variables; null for circuits streamed from the generator (no names)
  size_t off = 0, len = 0;  // its elements are vals_[off .. off+len) once it has been declared
  size_t io_offset = 0;  // signals: first element in the input / output vector
  int declared = 0;      // times the declaration has executed
};

struct U256Hash {
  size_t operator()(const U256& a) const {
    const uint64_t h = (a.w[0] ^ a.w[1] * 0xc2b2ae3d27d4eb4full ^ a.w[2] * 0x165667b19e3779f9ull ^ a.w[3]) * 0x9e3779b97f4a7c15ull;
    return (size_t)(h ^ h >> 29);
  }
};

class Pass1 {
 public:
  explicit Pass1(const LowerOptions& o) : opts_(o) {
    max_ops_ = o.max_ops ? o.max_ops : (1u << 22);
    max_iterations_ = o.max_iterations ? o.max_iterations : (1u << 20);
    // the arrays of the last circuit this thread lowered, emptied: their memory is mapped and warm
    Storage& st = storage();
    ops.swap(st.ops);
    stmt_begin.swap(st.stmt_begin);
    consts.swap(st.consts);
    cells_.swap(st.cells);
    vals_.swap(st.vals);
    const_table_.swap(st.const_table);
    ops.clear();
    stmt_begin.clear();
    consts.clear();
    cells_.clear();
    vals_.clear();
    const_table_.clear();
  }
  ~Pass1() {
    Storage& st = storage();
    if (ops.capacity() > (1u << 22)) return;  // a giant circuit's arrays are not kept
    ops.swap(st.ops);
    stmt_begin.swap(st.stmt_begin);
    consts.swap(st.consts);
    cells_.swap(st.cells);
    vals_.swap(st.vals);
    const_table_.swap(st.const_table);
  }
  Pass1(const Pass1&) = delete;
  Pass1& operator=(const Pass1&) = delete;

  std::vector<SOp> ops;
  std::vector<uint32_t> stmt_begin;  // first op of every executed statement that is not a block (its ops stay together when the tape is reordered)
  std::vector<U256> consts;
  int32_t n_values = 0;
  size_t n_inputs = 0, n_outputs = 0;
  std::vector<Tape::IoEntry> inputs, outputs;

  // A variable of the circuit, in declaration order.  Signal layout: declaration order, arrays flattened.
  // Returns its position (what the streaming calls below name it by).
  size_t add_variable(size_t id, VariableRole role, bool is_array, size_t type_len, const std::string* name) {
    Cells c;
    c.id = id;
    c.role = role;
    c.is_array = is_array;
    c.type_len = type_len;
    c.name = name;
    const size_t len = is_array ? type_len : 1;
    if (role.is_input()) {
      c.io_offset = n_inputs;
      inputs.push_back({id, n_inputs, len});
      n_inputs += len;
    } else if (role.is_output()) {
      c.io_offset = n_outputs;
      outputs.push_back({id, n_outputs, len});
      n_outputs += len;
      output_written_.resize(n_outputs, false);
    }
    cells_.push_back(c);
    return cells_.size() - 1;
  }

  void run(const Circuit& circuit) {
    cells_.reserve(circuit.variables.size());
    for (const Variable& v : circuit.variables) add_variable(v.var_ref.id, v.role, v.var_type.is_array, v.var_type.len, &v.name);
    // id -> position in cells_: a table when the ids are dense (the generator numbers them 0, 1, 2, ...), else a map
    size_t max_id = 0;
    for (const Variable& v : circuit.variables) max_id = std::max(max_id, v.var_ref.id);
    const bool dense = max_id < 4 * cells_.size() + 1024;
    if (dense) pos_of_id_.assign(max_id + 1, -1);
    for (size_t k = 0; k < cells_.size(); k++) {
      const size_t id = cells_[k].id;
      if (dense) {
        if (pos_of_id_[id] >= 0) fail(ZKIR_LOWER_BAD_AST, "duplicate variable id");
        pos_of_id_[id] = (int32_t)k;
      } else if (!pos_of_sparse_id_.emplace(id, (uint32_t)k).second) {
        fail(ZKIR_LOWER_BAD_AST, "duplicate variable id");
      }
    }
    dense_ids_ = dense;
    exec_block(circuit.instructions);
    finish();
  }

  // every output element is defined: the ones nothing constrained read as 0
  void finish() {
    stmt_begin.push_back((uint32_t)ops.size());
    for (size_t k = 0; k < n_outputs; k++)
      if (!output_written_[k]) emit({ZKIR_OP_OUTPUT, (Ref)k, const_of(U256()), kNoRef});
  }

  // ---- one statement at a time, for a caller that makes the circuit as it goes (variables by position) ----
  void stream_declare(size_t pos, const Expr* default_value) {
    note_statement();
    declare_cell(cells_[pos], default_value);
  }
  void stream_assign(size_t pos, const Expr& value) {
    note_statement();
    assign_cell(cells_[pos], value);
  }
  void stream_array_assign(size_t pos, const Expr& index, const Expr& value) {
    note_statement();
    array_assign_cell(cells_[pos], index, value);
  }
  void stream_constrain(size_t pos, const Expr& value) {
    note_statement();
    constrain_cell(cells_[pos], std::nullopt, value);
  }
  Ref constant(const U256& v) { return const_of(v); }  // the pool entry of a value (made if it is new)
  void stream_ids_are_positions() { ids_are_positions_ = true; }
  void reserve_for(size_t n_statements) {  // a circuit of about this many statements is coming
    cells_.reserve(n_statements + 32);
    vals_.reserve(n_statements * 3 + 64);
    ops.reserve(n_statements * 3 + 64);
    stmt_begin.reserve(n_statements + 8);
    consts.reserve(n_statements + 64);
    quads_.reserve(n_statements + 8);
  }

 private:
  [[noreturn]] void fail(int32_t cls, const std::string& m) { throw LowerError(cls, m); }
  static std::string name_of(const Cells& c) { return c.name ? *c.name : "#" + std::to_string(c.id); }
  void note_statement() {
    if (stmt_begin.empty() || stmt_begin.back() != ops.size()) stmt_begin.push_back((uint32_t)ops.size());
  }

  // the pool holds every value once: an open-addressing table of pool indices (most constants of a generated
  // circuit are fresh 64-bit draws, so a node-per-entry map spends its time allocating)
  Ref const_of(const U256& v) {
    if ((consts.size() + 1) * 2 > const_table_.size()) grow_const_table();
    const size_t mask = const_table_.size() - 1;
    size_t at = U256Hash()(v) & mask;
    for (;; at = (at + 1) & mask) {
      const uint32_t e = const_table_[at];
      if (e == UINT32_MAX) break;
      if (consts[e] == v) return const_ref(e);
    }
    const uint32_t id = (uint32_t)consts.size();
    consts.push_back(v);
    const_table_[at] = id;
    return const_ref(id);
  }
  void grow_const_table() {
    size_t cap = std::max<size_t>(64, const_table_.size() * 2);
    if (const_table_.empty())  // the table of the circuit before (memory kept): start at half its size, not from 64 again
      while (cap * 4 <= const_table_.capacity()) cap *= 2;
    while (cap < (consts.size() + 1) * 2) cap *= 2;
    const_table_.assign(cap, UINT32_MAX);
    for (uint32_t e = 0; e < consts.size(); e++) {
      size_t at = U256Hash()(consts[e]) & (cap - 1);
      while (const_table_[at] != UINT32_MAX) at = (at + 1) & (cap - 1);
      const_table_[at] = e;
    }
  }
  void emit(const SOp& op) {
    if (ops.size() >= max_ops_) fail(ZKIR_LOWER_LOOP_CAP, "tape longer than the op cap");
    ops.push_back(op);
  }
  Ref new_value(uint8_t code, Ref a, Ref b) {
    const Ref v = n_values++;
    emit({code, v, a, b});
    return v;
  }
  // the tape operand that holds `v`
  Ref ref_of(Val& v) {
    if (v.ref == kNoRef) v.ref = const_of(v.value);
    return v.ref;
  }
  static Val known_val(const U256& x) {
    Val v;
    v.assigned = true;
    v.known = true;
    v.value = x;
    return v;
  }

  Cells& cells_of(const VarRef& r) {
    if (ids_are_positions_) {
      if (r.id < cells_.size()) return cells_[r.id];
    } else if (dense_ids_) {
      if (r.id < pos_of_id_.size() && pos_of_id_[r.id] >= 0) return cells_[(size_t)pos_of_id_[r.id]];
    } else {
      auto it = pos_of_sparse_id_.find(r.id);
      if (it != pos_of_sparse_id_.end()) return cells_[it->second];
    }
    fail(ZKIR_LOWER_BAD_AST, "variable " + std::to_string(r.id) + " not found");
  }

  Val read_var(const VarRef& r) {
    Cells& c = cells_of(r);
    if (!c.declared) fail(ZKIR_LOWER_BAD_AST, "variable " + name_of(c) + " used before its declaration");
    size_t k = 0;
    if (c.is_array) {
      if (!r.idx) fail(ZKIR_LOWER_BAD_AST, "array " + name_of(c) + " referenced without an index");
      if (*r.idx >= c.len) fail(ZKIR_LOWER_OOB_INDEX, "index out of bounds on " + name_of(c));
      k = *r.idx;
    } else if (r.idx) {
      fail(ZKIR_LOWER_BAD_AST, "index on the scalar " + name_of(c));
    }
    Val& v = vals_[c.off + k];
    if (!v.assigned) fail(ZKIR_LOWER_UNINIT_SIGNAL, "signal " + name_of(c) + " read before it is assigned");
    if (c.role.is_input() && v.ref == kNoRef) {
      // first read of this input element: load it (and convert it) once
      const Ref x = n_values++;
      emit({ZKIR_OP_INPUT, x, (Ref)(c.io_offset + k), kNoRef});
      v.ref = x;
    }
    return v;
  }

  // ---- expressions ----
  static uint8_t opcode_of(ExprKind k, uint32_t bnot_variant) {
    switch (k) {
      case ExprKind::Add: return ZKIR_OP_ADD;
      case ExprKind::Sub: return ZKIR_OP_SUB;
      case ExprKind::Mul: return ZKIR_OP_MUL;
      case ExprKind::Power: return ZKIR_OP_POW;
      case ExprKind::Div: return ZKIR_OP_DIV;
      case ExprKind::IntDiv: return ZKIR_OP_IDIV;
      case ExprKind::Rem: return ZKIR_OP_REM;
      case ExprKind::LessThan: return ZKIR_OP_LT;
      case ExprKind::LessThanEq: return ZKIR_OP_LE;
      case ExprKind::GreaterThan: return ZKIR_OP_GT;
      case ExprKind::GreaterThanEq: return ZKIR_OP_GE;
      case ExprKind::Equal: return ZKIR_OP_EQ;
      case ExprKind::And: return ZKIR_OP_AND;
      case ExprKind::Or: return ZKIR_OP_OR;
      case ExprKind::BitAnd: return ZKIR_OP_BAND;
      case ExprKind::BitOr: return ZKIR_OP_BOR;
      case ExprKind::BitXor: return ZKIR_OP_BXOR;
      case ExprKind::BitRShift: return ZKIR_OP_SHR;
      case ExprKind::BitLShift: return ZKIR_OP_SHL;
      case ExprKind::BitNot: return bnot_variant ? ZKIR_OP_BNOT256 : ZKIR_OP_BNOT;
      default: return ZKIR_OP_NOP;
    }
  }

  U256 fold(ExprKind k, const U256& a, const U256& b) {
    const U256 one = U256::from_u64(1), zero;
    U256 r;
    bool ok = true;
    switch (k) {
      case ExprKind::Add: return fr::add(a, b);
      case ExprKind::Sub: return fr::sub(a, b);
      case ExprKind::Mul: return fr::mul(a, b);
      case ExprKind::Power: return fr::pow(a, b);
      case ExprKind::Div: ok = fr::div(a, b, &r); break;
      case ExprKind::IntDiv: ok = fr::idiv(a, b, &r); break;
      case ExprKind::Rem: ok = fr::rem(a, b, &r); break;
      case ExprKind::LessThan: return fr::lt(a, b) ? one : zero;
      case ExprKind::LessThanEq: return fr::lt(b, a) ? zero : one;
      case ExprKind::GreaterThan: return fr::lt(b, a) ? one : zero;
      case ExprKind::GreaterThanEq: return fr::lt(a, b) ? zero : one;
      case ExprKind::Equal: return a == b ? one : zero;
      case ExprKind::And: return (!a.is_zero() && !b.is_zero()) ? one : zero;
      case ExprKind::Or: return (!a.is_zero() || !b.is_zero()) ? one : zero;
      case ExprKind::BitAnd: return fr::band(a, b);
      case ExprKind::BitOr: return fr::bor(a, b);
      case ExprKind::BitXor: return fr::bxor(a, b);
      case ExprKind::BitRShift: return fr::shr(a, b);
      case ExprKind::BitLShift: return fr::shl(a, b);
      case ExprKind::BitNot: return opts_.bnot_variant ? fr::bnot256(a) : fr::bnot254(a);
      default: fail(ZKIR_LOWER_BAD_AST, "not an operator");
    }
    if (!ok) fail(ZKIR_LOWER_DIV_BY_ZERO, "division by a known zero");
    return r;
  }

  // degree bookkeeping: circom accepts `<==` only when the right-hand side can be
  // written A*B + C with A, B, C of degree <= 1 in the signals
  void combine(ExprKind k, Val& a, Val& b, Val* r) {
    const Degree da = a.deg, db = b.deg;
    r->deg = kNonQuadratic;
    if (k == ExprKind::Add || k == ExprKind::Sub) {
      if (da <= kLinear && db <= kLinear) {
        r->deg = kLinear;
      } else if ((da == kQuadratic && db <= kLinear) || (da <= kLinear && db == kQuadratic)) {
        Val& q = da == kQuadratic ? a : b;
        Val& l = da == kQuadratic ? b : a;
        r->deg = kQuadratic;
        // the linear operand as at most one term (one allocation for the new term list, none for the operand)
        const bool has_term = !(l.known && l.value.is_zero());
        Term lt{U256::from_u64(1), has_term ? ref_of(l) : kNoRef};
        const Quad& old = quads_[(size_t)q.quad];  // (ref_of above may add a constant, never a quad)
        Quad nq;
        nq.qa = old.qa;
        nq.qb = old.qb;
        nq.scale = old.scale;
        nq.terms.reserve(old.terms.size() + 1);
        nq.terms = old.terms;
        if (k == ExprKind::Sub) {
          if (&q == &b) {  // l - q: negate the quadratic part
            nq.scale = fr::neg(nq.scale);
            for (Term& t : nq.terms) t.coef = fr::neg(t.coef);
          } else {  // q - l
            lt.coef = fr::neg(lt.coef);
          }
        }
        if (has_term) nq.terms.push_back(lt);
        r->quad = (int32_t)quads_.size();
        quads_.push_back(std::move(nq));
      }
    } else if (k == ExprKind::Mul) {
      if (da == kKnown || db == kKnown) {
        Val& kn = da == kKnown ? a : b;
        Val& ot = da == kKnown ? b : a;
        if (ot.deg <= kLinear) {
          r->deg = ot.deg;  // known * linear stays linear
        } else if (ot.deg == kQuadratic) {
          r->deg = kQuadratic;
          Quad nq = quads_[(size_t)ot.quad];
          nq.scale = fr::mul(nq.scale, kn.value);
          for (Term& t : nq.terms) t.coef = fr::mul(t.coef, kn.value);
          r->quad = (int32_t)quads_.size();
          quads_.push_back(std::move(nq));
        }
      } else if (da == kLinear && db == kLinear) {
        r->deg = kQuadratic;
        Quad nq;
        nq.qa = ref_of(a);
        nq.qb = ref_of(b);
        nq.scale = U256::from_u64(1);
        r->quad = (int32_t)quads_.size();
        quads_.push_back(std::move(nq));
      }
    } else if (k == ExprKind::Div && db == kKnown) {
      // division by a known non-zero value is multiplication by its inverse
      U256 inv;
      if (!fr::div(U256::from_u64(1), b.value, &inv)) fail(ZKIR_LOWER_DIV_BY_ZERO, "division by a known zero");
      if (da <= kLinear) {
        r->deg = da;
      } else if (da == kQuadratic) {
        r->deg = kQuadratic;
        Quad nq = quads_[(size_t)a.quad];
        nq.scale = fr::mul(nq.scale, inv);
        for (Term& t : nq.terms) t.coef = fr::mul(t.coef, inv);
        r->quad = (int32_t)quads_.size();
        quads_.push_back(std::move(nq));
      }
    }
    if (r->deg == kKnown) r->deg = kLinear;  // not known: at least one operand depends on a signal
  }

  Val eval(const Expr& e) {
    if (++eval_depth_ > 2000) fail(ZKIR_LOWER_BAD_AST, "expression nested too deeply");
    struct Guard {
      int& d;
      ~Guard() { d--; }
    } guard{eval_depth_};
    switch (e.kind) {
      case ExprKind::Constant: {
        if (e.constant.is_u64()) return known_val(U256::from_u64(e.constant.u64()));
        U256 v;
        const std::string text = e.constant.str();
        if (!fr::parse_decimal(text, &v)) fail(ZKIR_LOWER_BAD_AST, "constant is not a decimal integer: " + text);
        return known_val(v);
      }
      case ExprKind::Var:
        return read_var(e.var);
      default:
        break;
    }
    if (!e.left || (e.is_binary() && !e.right)) fail(ZKIR_LOWER_BAD_AST, "operator without operands");
    Val a = eval(*e.left);
    Val b = e.is_binary() ? eval(*e.right) : known_val(U256());
    const uint8_t code = opcode_of(e.kind, opts_.bnot_variant);
    Val r;
    r.assigned = true;
    r.from_signal = a.from_signal || b.from_signal;
    if (a.known && b.known) {
      r.known = true;
      r.deg = kKnown;
      r.value = fold(e.kind, a.value, b.value);  // also raises division by a known zero
      if (!opts_.fold_constants) r.ref = new_value(code, ref_of(a), e.is_binary() ? ref_of(b) : kNoRef);
      return r;
    }
    if ((e.kind == ExprKind::Div || e.kind == ExprKind::IntDiv || e.kind == ExprKind::Rem) && b.known &&
        b.value.is_zero())
      fail(ZKIR_LOWER_DIV_BY_ZERO, "division by a known zero");
    combine(e.kind, a, b, &r);
    r.from_signal = a.from_signal || b.from_signal;
    r.ref = new_value(code, ref_of(a), e.is_binary() ? ref_of(b) : kNoRef);
    return r;
  }

  // ---- statements ----
  void declare(const Instruction& ins) { declare_cell(cells_of(ins.var.var_ref), ins.default_value); }
  void declare_cell(Cells& c, const Expr* default_value) {
    const size_t len = c.is_array ? c.type_len : 1;
    if (c.role.is_signal && c.declared) fail(ZKIR_LOWER_DOUBLE_ASSIGN, "signal " + name_of(c) + " declared again by a loop");
    c.declared++;
    if (c.len != len) {  // first execution of the declaration: its elements go at the end of vals_
      c.off = vals_.size();
      c.len = len;
      vals_.resize(vals_.size() + len);
    }
    Val* cell = vals_.data() + c.off;
    for (size_t k = 0; k < len; k++) cell[k] = Val();
    if (c.role.is_local()) {
      for (size_t k = 0; k < len; k++) cell[k] = known_val(U256());  // circom variables start at 0
    } else if (c.role.is_input()) {
      for (size_t k = 0; k < len; k++) {
        cell[k].assigned = true;
        cell[k].deg = kLinear;
        cell[k].from_signal = true;
      }
    }
    if (default_value) {
      if (!c.role.is_local() || c.is_array) fail(ZKIR_LOWER_BAD_AST, "default value on " + name_of(c));
      Val d = eval(*default_value);
      vals_[c.off] = std::move(d);
    }
  }

  void assign(const Instruction& ins) { assign_cell(cells_of(ins.target), *ins.value); }
  void assign_cell(Cells& c, const Expr& value) {
    if (!c.role.is_local() || c.is_array) fail(ZKIR_LOWER_BAD_AST, "Assign target must be a local field");
    if (!c.declared) fail(ZKIR_LOWER_BAD_AST, "assignment before declaration");
    Val v = eval(value);
    vals_[c.off] = std::move(v);
  }

  void array_assign(const Instruction& ins) { array_assign_cell(cells_of(ins.target), *ins.index, *ins.value); }
  void array_assign_cell(Cells& c, const Expr& index, const Expr& value) {
    if (!c.role.is_local() || !c.is_array) fail(ZKIR_LOWER_BAD_AST, "ArrayAssign target must be a local array");
    if (!c.declared) fail(ZKIR_LOWER_BAD_AST, "assignment before declaration");
    Val idx = eval(index);
    if (!idx.known || idx.from_signal) fail(ZKIR_LOWER_UNKNOWN_CONTROL, "array index depends on a signal");
    if (idx.value.w[1] | idx.value.w[2] | idx.value.w[3] || idx.value.w[0] >= c.len)
      fail(ZKIR_LOWER_OOB_INDEX, "index out of bounds on " + name_of(c));
    Val v = eval(value);
    vals_[c.off + (size_t)idx.value.w[0]] = std::move(v);
  }

  // sum(coef * ref) as one tape value; kNoRef when there are no terms
  Ref materialize_terms(const std::vector<Term>& terms) {
    Ref acc = kNoRef;
    const U256 one = U256::from_u64(1);
    for (const Term& t : terms) {
      if (t.coef.is_zero()) continue;
      Ref x = t.ref;
      if (!(t.coef == one)) x = new_value(ZKIR_OP_MUL, const_of(t.coef), x);
      acc = acc == kNoRef ? x : new_value(ZKIR_OP_ADD, acc, x);
    }
    return acc;
  }

  void constrain(const Instruction& ins) { constrain_cell(cells_of(ins.target), ins.target.idx, *ins.value); }
  void constrain_cell(Cells& c, const std::optional<size_t>& target_idx, const Expr& value) {
    if (!c.role.is_output()) fail(ZKIR_LOWER_BAD_AST, "Constraint target must be an output signal");
    if (!c.declared) fail(ZKIR_LOWER_BAD_AST, "constraint before declaration");
    size_t k = 0;
    if (c.is_array) {
      if (!target_idx) fail(ZKIR_LOWER_BAD_AST, "array signal constrained without an index");
      if (*target_idx >= c.len) fail(ZKIR_LOWER_OOB_INDEX, "index out of bounds on " + name_of(c));
      k = *target_idx;
    }
    if (vals_[c.off + k].assigned) fail(ZKIR_LOWER_DOUBLE_ASSIGN, "signal " + name_of(c) + " assigned twice");
    Val v = eval(value);
    if (v.deg == kNonQuadratic) fail(ZKIR_LOWER_NON_QUADRATIC, "non quadratic constraint on " + name_of(c));
    // `<==` is `<--` followed by `===`: the signal takes the value, then A*B == C is checked.  (The OUTPUT op stays
    // ahead of the CHECK: behind it the signal has to be kept for three ops longer, and what that adds in slot
    // traffic costs more than a kernel can save by reusing the operands its MUL unpacked: measured, +7 %.)
    const Ref out = ref_of(v);
    const size_t out_index = c.io_offset + k;
    emit({ZKIR_OP_OUTPUT, (Ref)out_index, out, kNoRef});
    output_written_[out_index] = true;
    const Ref one = const_of(U256::from_u64(1));
    if (v.deg == kQuadratic) {
      const Quad& Q = quads_[(size_t)v.quad];  // (the ops emitted below add values and constants, never a quad: the reference stays good)
      Ref A = Q.qa;
      if (!(Q.scale == U256::from_u64(1))) A = new_value(ZKIR_OP_MUL, const_of(Q.scale), A);
      const Ref lin = materialize_terms(Q.terms);
      const Ref C = lin == kNoRef ? out : new_value(ZKIR_OP_SUB, out, lin);
      emit({ZKIR_OP_CHECK, C, A, Q.qb});
    } else {
      emit({ZKIR_OP_CHECK, out, out, one});  // value * 1 == signal
    }
    // from here on the signal is a plain degree-1 quantity
    Val sig;
    sig.assigned = true;
    sig.known = v.known;
    sig.from_signal = true;
    sig.value = v.value;
    sig.ref = out;
    sig.deg = v.known ? kKnown : kLinear;
    vals_[c.off + k] = std::move(sig);
  }

  bool truth(const Expr& cond, const char* what) {
    Val v = eval(cond);
    if (!v.known || v.from_signal) fail(ZKIR_LOWER_UNKNOWN_CONTROL, std::string(what) + " condition depends on a signal");
    return !v.value.is_zero();
  }

  void exec(const Instruction& ins) {
    if (ins.kind != InstrKind::If && ins.kind != InstrKind::For && ins.kind != InstrKind::While) note_statement();
    switch (ins.kind) {
      case InstrKind::VarDecl: declare(ins); return;
      case InstrKind::Assign: assign(ins); return;
      case InstrKind::ArrayAssign: array_assign(ins); return;
      case InstrKind::Constraint: constrain(ins); return;
      case InstrKind::If:
        if (truth(*ins.cond, "if"))
          exec_block(ins.then_branch());
        else if (ins.else_branch())
          exec_block(*ins.else_branch());
        return;
      case InstrKind::For: {
        if (!ins.init() || !ins.step() || !ins.cond) fail(ZKIR_LOWER_BAD_AST, "malformed for loop");
        exec(*ins.init());
        for (;;) {
          if (!truth(*ins.cond, "loop")) break;
          tick();
          exec_block(ins.body());
          exec(*ins.step());
        }
        return;
      }
      case InstrKind::While:
        while (truth(*ins.cond, "loop")) {
          tick();
          exec_block(ins.body());
        }
        return;
    }
  }
  void exec_block(const std::vector<Instruction>& b) {
    if (++block_depth_ > 256) fail(ZKIR_LOWER_BAD_AST, "blocks nested too deeply");
    for (const Instruction& i : b) exec(i);
    block_depth_--;
  }
  // loop iterations are bounded independently of the ops they emit: a loop whose body
  // folds to nothing must still stop
  void tick() {
    if (++iterations_ > max_iterations_) fail(ZKIR_LOWER_LOOP_CAP, "loop trip cap exceeded");
  }

  struct Storage {
    std::vector<SOp> ops;
    std::vector<uint32_t> stmt_begin, const_table;
    std::vector<U256> consts;
    std::vector<Cells> cells;
    std::vector<Val> vals;
  };
  static Storage& storage() {
    static thread_local Storage st;
    return st;
  }
  LowerOptions opts_;
  bool ids_are_positions_ = false;
  size_t max_ops_;
  uint64_t max_iterations_;
  std::vector<Cells> cells_;  // one per circuit variable, in declaration order
  std::vector<Val> vals_;     // the elements of every declared variable
  std::vector<Quad> quads_;   // what Val::quad points to
  std::vector<int32_t> pos_of_id_;
  std::unordered_map<size_t, uint32_t> pos_of_sparse_id_;
  bool dense_ids_ = true;
  std::vector<uint32_t> const_table_;  // open addressing: pool index or UINT32_MAX
  std::vector<bool> output_written_;
  uint64_t iterations_ = 0;
  int eval_depth_ = 0, block_depth_ = 0;
};

// the ops whose `a` is an input element / whose `dst` is an output element, not a value
inline bool is_input_op(uint8_t code) { return code == ZKIR_OP_INPUT || code == ZKIR_OP_INPUT_SCALED; }
inline bool is_output_op(uint8_t code) { return code == ZKIR_OP_OUTPUT || code == ZKIR_OP_OUTPUT_SCALED; }

// ---- pass 2: homes for values ----
struct Home {
  uint32_t operand = ZKIR_OPERAND_NONE;  // GSLOT / LSLOT operand, or NONE when the value only lives in the accumulator
};

// One op stream (the whole tape, or one strand of a strand tape) gets its accumulator
// forwarding and its private slots here.  `xhome[v]` is the exchange-slot operand of a value
// that other strands read (strand tapes only; ZKIR_OPERAND_NONE otherwise); `check_id[i]` is the
// constraint number of CHECK op i of `all`, written into the op when `carry_ids`.
struct StreamOut {
  std::vector<zkir_op> ops;
  uint32_t n_g = 0, used_l = 0, n_mul = 0;
};

StreamOut place_stream_cached(const std::vector<SOp>& all, const std::vector<uint32_t>& stream, int32_t nv,
                              const std::vector<uint32_t>& check_id, bool carry_ids, const LowerOptions& opts,
                              std::vector<uint32_t>* triples);

StreamOut place_stream(const std::vector<SOp>& all, const std::vector<uint32_t>& stream, int32_t nv,
                       const std::vector<uint32_t>& xhome, const std::vector<uint32_t>& check_id, bool carry_ids,
                       const LowerOptions& opts, std::vector<uint32_t>* triples) {
  const size_t n = stream.size();
  auto op_at = [&](size_t i) -> const SOp& { return all[stream[i]]; };
  auto defines = [](const SOp& o) {
    return !is_output_op(o.code) && o.code != ZKIR_OP_CHECK && o.code != ZKIR_OP_NOP && o.code != ZKIR_OP_BARRIER;
  };
  auto is_x = [&](Ref r) { return !xhome.empty() && (xhome[r] >> ZKIR_KIND_SHIFT) == ZKIR_KIND_XSLOT; };

  // last op that reads each value, and whether every read can come from the accumulator:
  // the accumulator holds value v from its defining op until the next defining op has run
  static thread_local struct {  // (memory kept from call to call, per thread: fresh arrays per circuit are mostly page faults)
    std::vector<int64_t> def_at, last_use, next_def;
    std::vector<Home> home;
    std::vector<int32_t> dies_head, dies_next;
  } W;
  std::vector<int64_t>&def_at = W.def_at, &last_use = W.last_use;
  def_at.assign(nv, -1);
  last_use.assign(nv, -1);
  std::vector<int64_t>& next_def = W.next_def;  // index of the first defining op at or after i
  next_def.assign(n + 1, (int64_t)n);
  for (int64_t i = (int64_t)n - 1; i >= 0; i--) next_def[i] = defines(op_at(i)) ? i : next_def[i + 1];
  auto note_use = [&](Ref r, int64_t at) {
    if (r >= 0) last_use[r] = std::max(last_use[r], at);
  };
  for (size_t i = 0; i < n; i++) {
    const SOp& o = op_at(i);
    if (o.code == ZKIR_OP_BARRIER) continue;
    if (is_input_op(o.code)) {
      def_at[o.dst] = (int64_t)i;
      continue;
    }
    note_use(o.a, (int64_t)i);
    note_use(o.b, (int64_t)i);
    if (o.code == ZKIR_OP_CHECK) note_use(o.dst, (int64_t)i);
    if (defines(o)) def_at[o.dst] = (int64_t)i;
  }

  std::vector<Home>& home = W.home;
  home.assign(nv, Home());
  std::vector<uint32_t> free_l, free_g;
  for (uint32_t k = opts.n_lslots; k-- > 0;) free_l.push_back(k);
  // values whose slot is released after op i
  std::vector<int32_t>&dies_head = W.dies_head, &dies_next = W.dies_next;  // per op: list of the values whose last read it is
  dies_head.assign(n, -1);
  dies_next.assign(nv, -1);
  const int64_t kWindowLifetime = opts.window_lifetime ? opts.window_lifetime : 48;

  StreamOut out;
  out.ops.resize(n);

  auto src_operand = [&](Ref r, size_t at) -> uint32_t {
    if (r == kNoRef) return ZKIR_OPERAND_NONE;
    if (is_const(r)) return ZKIR_OPERAND(ZKIR_KIND_CONST, const_index(r));
    // in the accumulator iff this stream defined it and no defining op lies strictly between
    // its definition and this use
    if (opts.use_acc && def_at[r] >= 0 && next_def[def_at[r] + 1] >= (int64_t)at) return ZKIR_OPERAND_ACC;
    if (is_x(r)) return xhome[r];
    if (def_at[r] < 0 || (home[r].operand >> ZKIR_KIND_SHIFT) == ZKIR_KIND_NONE)
      throw LowerError(ZKIR_LOWER_BAD_AST, "internal: value has no home");
    return home[r].operand;
  };

  for (size_t i = 0; i < n; i++) {
    const SOp& o = op_at(i);
    zkir_op t{};
    t.code = o.code;
    if (o.code == ZKIR_OP_BARRIER) {
      t.dst = t.a = t.b = ZKIR_OPERAND_NONE;
      out.ops[i] = t;
      continue;
    }
    if (is_input_op(o.code)) {
      t.a = ZKIR_OPERAND(ZKIR_KIND_INPUT, (uint32_t)o.a);
      t.b = ZKIR_OPERAND_NONE;
    } else {
      t.a = src_operand(o.a, i);
      t.b = src_operand(o.b, i);
    }
    // commutative ops take the accumulator as their first operand: the kernel then
    // multiplies / adds in place instead of copying it
    if ((o.code == ZKIR_OP_MUL || o.code == ZKIR_OP_ADD || o.code == ZKIR_OP_CHECK) &&
        (t.b >> ZKIR_KIND_SHIFT) == ZKIR_KIND_ACC && (t.a >> ZKIR_KIND_SHIFT) != ZKIR_KIND_ACC)
      std::swap(t.a, t.b);
    if (is_output_op(o.code)) {
      t.dst = (uint32_t)o.dst;
    } else if (o.code == ZKIR_OP_CHECK) {
      t.dst = src_operand(o.dst, i);
      const uint32_t id = check_id[stream[i]];
      if (carry_ids) t.code |= (id + 1) << 8;
      if (triples) {
        (*triples)[3 * (size_t)id] = t.a;
        (*triples)[3 * (size_t)id + 1] = t.b;
        (*triples)[3 * (size_t)id + 2] = t.dst;
      }
    } else {
      if (o.code == ZKIR_OP_MUL) out.n_mul++;
      const int32_t v = o.dst;
      // does any read fall outside the accumulator's reach?
      const bool acc_only = opts.use_acc && last_use[v] < next_def[i + 1] + 1 && last_use[v] <= next_def[i + 1];
      if (is_x(v)) {
        t.dst = xhome[v];  // other strands read it: it lives in its exchange slot, for this strand too
      } else if (last_use[v] < 0 || acc_only) {
        t.dst = ZKIR_OPERAND_NONE;
      } else {
        const bool window_ok = !free_l.empty() && (last_use[v] - (int64_t)i) <= kWindowLifetime;
        if (window_ok) {
          const uint32_t s = free_l.back();
          free_l.pop_back();
          out.used_l = std::max(out.used_l, s + 1);
          home[v].operand = ZKIR_OPERAND(ZKIR_KIND_LSLOT, s);
        } else {
          uint32_t s;
          if (!free_g.empty() && opts.reuse_slots) {
            s = free_g.back();
            free_g.pop_back();
          } else {
            s = out.n_g++;
          }
          home[v].operand = ZKIR_OPERAND(ZKIR_KIND_GSLOT, s);
        }
        t.dst = home[v].operand;
        if (opts.reuse_slots) {
          dies_next[v] = dies_head[(size_t)last_use[v]];
          dies_head[(size_t)last_use[v]] = v;
        }
      }
    }
    out.ops[i] = t;
    // release the slots of values read for the last time by this op; a slot freed here can
    // be the destination of a LATER op only, never of this one (its store is already placed)
    for (int32_t v = dies_head[i]; v >= 0; v = dies_next[v]) {
      const uint32_t kind = home[v].operand >> ZKIR_KIND_SHIFT, idx = home[v].operand & ZKIR_INDEX_MASK;
      if (kind == ZKIR_KIND_LSLOT) free_l.push_back(idx);
      if (kind == ZKIR_KIND_GSLOT) free_g.push_back(idx);
    }
  }
  return out;
}

// The same for a plain tape, with the on-chip window run as a cache whose future is known.  A value takes a window
// slot at its definition when its next read comes sooner than that of the value it would push out (the resident whose
// next read lies farthest away goes); it is read from the window for as long as it stays there.  A value that is
// read again after it left the window also gets a device slot, written by a COPY op right behind its definition
// (the accumulator still holds it).  A circuit whose values are read at random points of a long program keeps, in
// the few slots the window has, the values about to be read, instead of only those whose every read is near.
StreamOut place_stream_cached(const std::vector<SOp>& all, const std::vector<uint32_t>& stream, int32_t nv,
                              const std::vector<uint32_t>& check_id, bool carry_ids, const LowerOptions& opts,
                              std::vector<uint32_t>* triples) {
  const size_t n = stream.size();
  auto op_at = [&](size_t i) -> const SOp& { return all[stream[i]]; };
  auto defines = [](const SOp& o) {
    return !is_output_op(o.code) && o.code != ZKIR_OP_CHECK && o.code != ZKIR_OP_NOP && o.code != ZKIR_OP_BARRIER;
  };
  // The arrays of this function keep their memory from call to call (per thread): a campaign lowers thousands of
  // circuits per second and thread, and fresh megabytes per circuit are mostly page faults.
  static thread_local struct {
    std::vector<int64_t> def_at, next_def, resident_until;
    std::vector<uint32_t> n_uses, use_pos, fill, cursor, lslot, gslot;
    std::vector<int32_t> l_dies_head, l_dies_next, g_dies_head, g_dies_next;
  } W;
  std::vector<int64_t>& def_at = W.def_at;
  def_at.assign(nv, -1);
  std::vector<int64_t>& next_def = W.next_def;
  next_def.assign(n + 2, (int64_t)n);
  for (int64_t i = (int64_t)n - 1; i >= 0; i--) next_def[i] = defines(op_at(i)) ? i : next_def[i + 1];
  // reads of every value that the accumulator cannot serve, in stream order (compressed rows)
  std::vector<uint32_t>& n_uses = W.n_uses;
  n_uses.assign(nv + 1, 0);
  auto sources = [&](const SOp& o, Ref out[3]) {
    out[0] = out[1] = out[2] = kNoRef;
    if (is_input_op(o.code) || o.code == ZKIR_OP_NOP || o.code == ZKIR_OP_BARRIER) return;
    out[0] = o.a;
    out[1] = o.b;
    if (o.code == ZKIR_OP_CHECK) out[2] = o.dst;
    if (out[1] == out[0]) out[1] = kNoRef;
    if (out[2] == out[0] || out[2] == out[1]) out[2] = kNoRef;
  };
  for (size_t i = 0; i < n; i++)
    if (defines(op_at(i)) && op_at(i).dst >= 0) def_at[op_at(i).dst] = (int64_t)i;
  auto in_acc = [&](Ref r, size_t at) { return opts.use_acc && def_at[r] >= 0 && next_def[def_at[r] + 1] >= (int64_t)at; };
  for (size_t i = 0; i < n; i++) {
    Ref src[3];
    sources(op_at(i), src);
    for (Ref r : src)
      if (r >= 0) {
        if (def_at[r] < 0 || def_at[r] >= (int64_t)i) throw LowerError(ZKIR_LOWER_BAD_AST, "internal: value read before its definition");
        if (!in_acc(r, i)) n_uses[r + 1]++;
      }
  }
  for (int32_t v = 0; v < nv; v++) n_uses[v + 1] += n_uses[v];
  std::vector<uint32_t>&use_pos = W.use_pos, &fill = W.fill;
  use_pos.assign(n_uses[nv], 0);
  fill.assign(n_uses.begin(), n_uses.end() - 1);
  for (size_t i = 0; i < n; i++) {
    Ref src[3];
    sources(op_at(i), src);
    for (Ref r : src)
      if (r >= 0 && !in_acc(r, i)) use_pos[fill[r]++] = (uint32_t)i;
  }
  std::vector<uint32_t>& cursor = W.cursor;  // per value: its first read not yet passed
  cursor.assign(n_uses.begin(), n_uses.end() - 1);
  const uint32_t kNever = UINT32_MAX;
  auto next_use = [&](int32_t v, size_t from) -> uint32_t {  // first read at or after `from` (calls come with growing `from`)
    uint32_t& c = cursor[v];
    while (c < n_uses[v + 1] && use_pos[c] < from) c++;
    return c < n_uses[v + 1] ? use_pos[c] : kNever;
  };

  // first walk: how long every value stays in the window (the last stream position it may be read there)
  std::vector<int64_t>& resident_until = W.resident_until;
  resident_until.assign(nv, -1);
  std::vector<int32_t> resident;
  const size_t window = std::min<size_t>(opts.n_lslots, ZKIR_MAX_LSLOTS);
  for (size_t i = 0; i < n && window; i++) {
    const SOp& o = op_at(i);
    if (!defines(o) || o.dst < 0) continue;
    const int32_t v = o.dst;
    for (size_t k = 0; k < resident.size();) {  // the dead leave
      if (next_use(resident[k], i) == kNever) {
        resident_until[resident[k]] = (int64_t)i - 1;
        resident[k] = resident.back();
        resident.pop_back();
      } else {
        k++;
      }
    }
    const uint32_t nu = next_use(v, i + 1);
    if (nu == kNever) continue;
    if (resident.size() < window) {
      resident.push_back(v);
      continue;
    }
    size_t far = SIZE_MAX;
    uint32_t far_use = nu;
    for (size_t k = 0; k < resident.size(); k++) {
      const uint32_t u = next_use(resident[k], i);
      if (u > far_use && u != (uint32_t)i) {  // (a value this very op reads keeps its slot: the op's result would land on it)
        far = k;
        far_use = u;
      }
    }
    if (far != SIZE_MAX) {
      resident_until[resident[far]] = (int64_t)i - 1;
      resident[far] = v;
    }
  }
  for (int32_t v : resident) resident_until[v] = (int64_t)n;
  // `resident_until[v] >= def` now means: v is in the window from its definition to that position

  // second walk: slots and ops
  std::fill(cursor.begin(), cursor.end(), 0);
  for (int32_t v = 0; v < nv; v++) cursor[v] = n_uses[v];
  std::vector<uint32_t>&lslot = W.lslot, &gslot = W.gslot;
  lslot.assign(nv, UINT32_MAX);
  gslot.assign(nv, UINT32_MAX);
  std::vector<uint32_t> free_l, free_g;
  for (uint32_t k = (uint32_t)window; k-- > 0;) free_l.push_back(k);
  std::vector<int32_t>&l_dies_head = W.l_dies_head, &l_dies_next = W.l_dies_next, &g_dies_head = W.g_dies_head, &g_dies_next = W.g_dies_next;
  l_dies_head.assign(n + 1, -1);
  l_dies_next.assign(nv, -1);
  g_dies_head.assign(n + 1, -1);
  g_dies_next.assign(nv, -1);
  StreamOut out;
  out.ops.reserve(n + n / 8);
  auto src_operand = [&](Ref r, size_t at) -> uint32_t {
    if (r == kNoRef) return ZKIR_OPERAND_NONE;
    if (is_const(r)) return ZKIR_OPERAND(ZKIR_KIND_CONST, const_index(r));
    if (in_acc(r, at)) return ZKIR_OPERAND_ACC;
    if (lslot[r] != UINT32_MAX && (int64_t)at <= resident_until[r]) return ZKIR_OPERAND(ZKIR_KIND_LSLOT, lslot[r]);
    if (gslot[r] == UINT32_MAX) throw LowerError(ZKIR_LOWER_BAD_AST, "internal: value has no home");
    return ZKIR_OPERAND(ZKIR_KIND_GSLOT, gslot[r]);
  };
  for (size_t i = 0; i < n; i++) {
    const SOp& o = op_at(i);
    zkir_op t{};
    t.code = o.code;
    bool also_device = false;
    uint32_t device_home = 0;
    if (is_input_op(o.code)) {
      t.a = ZKIR_OPERAND(ZKIR_KIND_INPUT, (uint32_t)o.a);
      t.b = ZKIR_OPERAND_NONE;
    } else {
      t.a = src_operand(o.a, i);
      t.b = src_operand(o.b, i);
    }
    if ((o.code == ZKIR_OP_MUL || o.code == ZKIR_OP_ADD || o.code == ZKIR_OP_CHECK) &&
        (t.b >> ZKIR_KIND_SHIFT) == ZKIR_KIND_ACC && (t.a >> ZKIR_KIND_SHIFT) != ZKIR_KIND_ACC)
      std::swap(t.a, t.b);
    if (is_output_op(o.code)) {
      t.dst = (uint32_t)o.dst;
    } else if (o.code == ZKIR_OP_CHECK) {
      t.dst = src_operand(o.dst, i);
      const uint32_t id = check_id[stream[i]];
      if (carry_ids) t.code |= (id + 1) << 8;
      if (triples) {
        (*triples)[3 * (size_t)id] = t.a;
        (*triples)[3 * (size_t)id + 1] = t.b;
        (*triples)[3 * (size_t)id + 2] = t.dst;
      }
    } else if (o.code == ZKIR_OP_NOP || o.code == ZKIR_OP_BARRIER) {
      t.dst = t.a = t.b = ZKIR_OPERAND_NONE;
    } else {
      if (o.code == ZKIR_OP_MUL) out.n_mul++;
      const int32_t v = o.dst;
      const uint32_t first = n_uses[v], end = n_uses[v + 1];  // its reads beyond the accumulator's reach
      t.dst = ZKIR_OPERAND_NONE;
      if (first < end) {
        const int64_t until = resident_until[v];
        uint32_t last_in_window = UINT32_MAX, n_after = 0;
        for (uint32_t c = first; c < end; c++) {
          if ((int64_t)use_pos[c] <= until) last_in_window = use_pos[c];
          else n_after++;
        }
        const uint32_t last = use_pos[end - 1];
        if (last_in_window != UINT32_MAX) {
          if (free_l.empty()) throw LowerError(ZKIR_LOWER_BAD_AST, "internal: window slot accounting");
          lslot[v] = free_l.back();
          free_l.pop_back();
          out.used_l = std::max(out.used_l, lslot[v] + 1);
          t.dst = ZKIR_OPERAND(ZKIR_KIND_LSLOT, lslot[v]);
          // the slot is free again once the value has left the window: after its last read there, or when its
          // successor arrives (position until + 1 defines it), whichever the first walk decided
          const size_t release = (size_t)std::min<int64_t>(until, (int64_t)n - 1);
          l_dies_next[v] = l_dies_head[release];
          l_dies_head[release] = v;
        }
        if (n_after || last_in_window == UINT32_MAX) {
          uint32_t g;
          if (!free_g.empty() && opts.reuse_slots) {
            g = free_g.back();
            free_g.pop_back();
          } else {
            g = out.n_g++;
          }
          gslot[v] = g;
          if (last_in_window == UINT32_MAX) {
            t.dst = ZKIR_OPERAND(ZKIR_KIND_GSLOT, g);
          } else {
            also_device = true;
            device_home = ZKIR_OPERAND(ZKIR_KIND_GSLOT, g);
          }
          if (opts.reuse_slots) {
            g_dies_next[v] = g_dies_head[last];
            g_dies_head[last] = v;
          }
        }
      }
    }
    out.ops.push_back(t);
    if (also_device) out.ops.push_back(zkir_op{ZKIR_OP_COPY, device_home, ZKIR_OPERAND_ACC, ZKIR_OPERAND_NONE});
    for (int32_t v = l_dies_head[i]; v >= 0; v = l_dies_next[v]) free_l.push_back(lslot[v]);
    for (int32_t v = g_dies_head[i]; v >= 0; v = g_dies_next[v]) free_g.push_back(gslot[v]);
  }
  return out;
}

// Device-slot reads and writes of a tape per input vector, as the generated-stream kernels perform them: their operand
// registers remember which operand word they hold (a and b), a device-slot value pushed out of them waits in one
// victim row of LDS, a store to a slot makes every copy of it stale (zkir_interp_lean_kernel, tools/gen_interp_asm.py).
uint64_t device_slot_traffic(const std::vector<zkir_op>& ops, uint32_t one_word) {
  uint32_t tag_a = UINT32_MAX, tag_b = UINT32_MAX, tag_v = UINT32_MAX;
  uint64_t n = 0;
  auto fetch = [&](uint32_t word, uint32_t* tag) {
    const uint32_t kind = word >> ZKIR_KIND_SHIFT;
    if (tag && word == *tag) return;
    if (tag && word == tag_v) {
      *tag = word;
      return;
    }
    if (kind == ZKIR_KIND_ACC) {
      if (tag == &tag_b) *tag = UINT32_MAX;
      return;
    }
    if (tag && (*tag >> ZKIR_KIND_SHIFT) == ZKIR_KIND_GSLOT) tag_v = *tag;
    if (kind == ZKIR_KIND_GSLOT) n++;
    if (tag) *tag = (kind == ZKIR_KIND_GSLOT || kind == ZKIR_KIND_LSLOT || kind == ZKIR_KIND_CONST) ? word : UINT32_MAX;
  };
  for (const zkir_op& op : ops) {
    const uint32_t code = op.code & 0xffu;
    if (code == ZKIR_OP_NOP || code == ZKIR_OP_BARRIER) continue;
    if (code == ZKIR_OP_INPUT_SCALED) {
      // loaded straight into the accumulator: the operand registers keep what they hold
    } else if (code == ZKIR_OP_INPUT) {
      tag_a = tag_b = UINT32_MAX;
    } else if (code == ZKIR_OP_OUTPUT || code == ZKIR_OP_OUTPUT_SCALED) {
      fetch(op.a, &tag_a);
      continue;
    } else if (code == ZKIR_OP_COPY) {
      fetch(op.a, nullptr);
    } else if (code == ZKIR_OP_NEG) {
      fetch(op.a, &tag_b);
      tag_a = UINT32_MAX;
    } else if (code == ZKIR_OP_CHECK) {
      if (op.b != one_word) fetch(op.b, &tag_b);  // against the constant 1 nothing is multiplied and no b is fetched
      fetch(op.a, &tag_a);                       // (the product goes to scratch registers: a and b keep their values)
      fetch(op.dst, &tag_b);
      continue;
    } else {
      fetch(op.b, &tag_b);
      fetch(op.a, &tag_a);
    }
    if (tag_a == op.dst) tag_a = UINT32_MAX;
    if (tag_b == op.dst) tag_b = UINT32_MAX;
    if (tag_v == op.dst) tag_v = UINT32_MAX;
    if ((op.dst >> ZKIR_KIND_SHIFT) == ZKIR_KIND_GSLOT) n++;
  }
  return n;
}

// ---- the order of the statements of a plain tape ----
// The ops of one statement stay together and in order; the statements themselves may run in any order that defines
// every value before it is read (ops are pure, single assignment).  A circuit whose constraints read earlier signals
// at random leaves that freedom wide open, and the device-memory traffic of its tape depends on it: a statement that
// reads what the operand registers, the victim row or the window still hold loads nothing for it.  Greedy: after
// each statement, among the ready statements that read one of the values the last statements read or defined, take
// the one that reads most of them (a value read for the last time counts more: it then never needs a device slot);
// with no such statement, the earliest ready one in program order.
std::vector<uint32_t> schedule_statements(const std::vector<SOp>& ops, const std::vector<uint32_t>& stmt_begin, int32_t nv,
                                          size_t lookahead) {
  const size_t n = ops.size();
  std::vector<uint32_t> begin;
  for (uint32_t b : stmt_begin)
    if (b < n && (begin.empty() || b > begin.back())) begin.push_back(b);
  if (begin.empty() || begin[0] != 0) begin.insert(begin.begin(), 0u);
  const size_t nb = begin.size();
  begin.push_back((uint32_t)n);
  std::vector<uint32_t> order(n);
  for (size_t i = 0; i < n; i++) order[i] = (uint32_t)i;
  if (nb < 3) return order;
  auto defines = [](const SOp& o) { return !is_output_op(o.code) && o.code != ZKIR_OP_CHECK && o.code != ZKIR_OP_NOP; };
  std::vector<int32_t> def_stmt(nv, -1);
  for (size_t b = 0; b < nb; b++)
    for (uint32_t i = begin[b]; i < begin[b + 1]; i++)
      if (defines(ops[i]) && ops[i].dst >= 0) def_stmt[ops[i].dst] = (int32_t)b;
  // values a statement reads that another statement defines (at most a few), and who reads each value
  std::vector<uint32_t> ext_begin(nb + 1, 0), ext;
  std::vector<uint32_t> n_readers(nv, 0), n_wait(nb, 0);
  for (size_t b = 0; b < nb; b++) {
    ext_begin[b] = (uint32_t)ext.size();
    for (uint32_t i = begin[b]; i < begin[b + 1]; i++) {
      const SOp& o = ops[i];
      if (is_input_op(o.code) || o.code == ZKIR_OP_NOP) continue;
      const Ref src[3] = {o.b, o.a, o.code == ZKIR_OP_CHECK ? o.dst : kNoRef};
      for (Ref r : src) {
        if (r < 0 || def_stmt[r] < 0 || def_stmt[r] == (int32_t)b) continue;
        bool seen = false;
        for (uint32_t q = ext_begin[b]; q < ext.size(); q++) seen = seen || ext[q] == (uint32_t)r;
        if (!seen) ext.push_back((uint32_t)r);
      }
    }
  }
  ext_begin[nb] = (uint32_t)ext.size();
  std::vector<uint32_t> readers_begin(nv + 1, 0);
  for (uint32_t r : ext) readers_begin[r + 1]++;
  for (int32_t v = 0; v < nv; v++) readers_begin[v + 1] += readers_begin[v];
  std::vector<uint32_t> readers(ext.size()), rfill(readers_begin.begin(), readers_begin.end() - 1);
  std::vector<uint32_t> succ_begin(nb + 1, 0);
  for (size_t b = 0; b < nb; b++) {
    // distinct defining statements this one waits for
    for (uint32_t q = ext_begin[b]; q < ext_begin[b + 1]; q++) {
      readers[rfill[ext[q]]++] = (uint32_t)b;
      n_readers[ext[q]]++;
      bool seen = false;
      for (uint32_t p = ext_begin[b]; p < q; p++) seen = seen || def_stmt[ext[p]] == def_stmt[ext[q]];
      if (!seen) {
        n_wait[b]++;
        succ_begin[(size_t)def_stmt[ext[q]] + 1]++;
      }
    }
  }
  for (size_t b = 0; b < nb; b++) succ_begin[b + 1] += succ_begin[b];
  std::vector<uint32_t> succ(succ_begin[nb]), sfill(succ_begin.begin(), succ_begin.end() - 1);
  for (size_t b = 0; b < nb; b++)
    for (uint32_t q = ext_begin[b]; q < ext_begin[b + 1]; q++) {
      bool seen = false;
      for (uint32_t p = ext_begin[b]; p < q; p++) seen = seen || def_stmt[ext[p]] == def_stmt[ext[q]];
      if (!seen) succ[sfill[def_stmt[ext[q]]]++] = (uint32_t)b;
    }

  std::priority_queue<uint32_t, std::vector<uint32_t>, std::greater<uint32_t>> ready;  // earliest first
  std::vector<uint8_t> state(nb, 0);  // 0 waiting, 1 ready, 2 done
  for (size_t b = 0; b < nb; b++)
    if (n_wait[b] == 0) {
      ready.push((uint32_t)b);
      state[b] = 1;
    }
  // the values within reach: what the last statement read (operand registers, victim row) and what the last few
  // statements defined (still cheap to keep in the window)
  // weights found by a search over generated circuits of the benchmark's kind (five seeds): reading the last reader's
  // operands counts little, closing a value defined among the last 16 counts most
  const size_t kRecent = 16;
  const float w_cache = 0.5f, w_recent = 1.f, w_last = 16.f;
  std::vector<uint32_t> recent;  // ring of value ids
  size_t recent_at = 0;
  std::vector<uint32_t> last_read;
  std::vector<uint32_t> stamp(nv, 0), in_recent(nv, 0), score_stamp(nb, 0);
  std::vector<float> score(nb, 0.f);
  uint32_t round = 0;
  std::vector<uint32_t> out;
  out.reserve(n);
  std::vector<uint32_t> touched;
  // A statement may run ahead of program order by a bounded number of statements only: every statement that is passed
  // over keeps its operands alive for longer, and a circuit whose values die young in program order must not be
  // turned into one that spills.
  size_t earliest = 0;  // the first statement not yet placed
  for (size_t placed = 0; placed < nb; placed++) {
    round++;
    while (earliest < nb && state[earliest] == 2) earliest++;
    // candidates: ready readers of the values within reach
    touched.clear();
    auto consider = [&](uint32_t v, float weight, bool is_recent) {
      if (stamp[v] == round) return;
      stamp[v] = round;
      const bool closes = is_recent && n_readers[v] == 1;  // its last reader: the value then never leaves the chip
      for (uint32_t q = readers_begin[v]; q < readers_begin[v + 1]; q++) {
        const uint32_t b = readers[q];
        if (state[b] != 1 || b - earliest > lookahead) continue;
        if (score_stamp[b] != round) {
          score_stamp[b] = round;
          score[b] = 0.f;
          touched.push_back(b);
        }
        score[b] += weight + (closes ? w_last : 0.f);
      }
    };
    for (uint32_t v : recent) in_recent[v] = round;
    for (uint32_t v : last_read) consider(v, w_cache + (in_recent[v] == round ? w_recent : 0.f), in_recent[v] == round);
    for (uint32_t v : recent) consider(v, w_recent, true);
    uint32_t best = UINT32_MAX;
    for (uint32_t b : touched)
      if (best == UINT32_MAX || score[b] > score[best] || (score[b] == score[best] && b < best)) best = b;
    if (best == UINT32_MAX) {
      while (!ready.empty() && state[ready.top()] != 1) ready.pop();
      if (ready.empty()) throw LowerError(ZKIR_LOWER_BAD_AST, "internal: statement scheduling made no progress");
      best = ready.top();
      ready.pop();
    }
    state[best] = 2;
    for (uint32_t i = begin[best]; i < begin[best + 1]; i++) {
      out.push_back(i);
      if (defines(ops[i]) && ops[i].dst >= 0) {
        if (recent.size() < kRecent) recent.push_back((uint32_t)ops[i].dst);
        else recent[recent_at++ % kRecent] = (uint32_t)ops[i].dst;
      }
    }
    last_read.assign(ext.begin() + ext_begin[best], ext.begin() + std::min<size_t>(ext_begin[best] + 3, ext_begin[best + 1]));
    for (uint32_t q = ext_begin[best]; q < ext_begin[best + 1]; q++) n_readers[ext[q]]--;
    for (uint32_t q = succ_begin[best]; q < succ_begin[best + 1]; q++)
      if (--n_wait[succ[q]] == 0) {
        state[succ[q]] = 1;
        ready.push(succ[q]);
      }
  }
  return out;
}

// ---- the order of the ops within one statement ----
// Pass 1 emits the ops of a statement in the order the expression is written: for `out <== A*B + C` that is A, B,
// the product, C, the sum.  The product then has to be kept somewhere while C is computed, next to A and B (the
// CHECK reads them again) and C (the CHECK's right-hand side subtracts it): four temporaries.  With C first the
// product goes from the multiplier straight into the sum (the accumulator forwards a value to the op after its
// definition) and three temporaries remain.  Every temporary beyond the rows of the LDS window costs a store to and
// a load from device memory per vector.  For each statement the order "operands of an op: the shallower chain first"
// (depth-first from the ops nothing in the statement reads) is compared with the order as written: the one with the
// fewest values alive at once that are not forwarded is kept, then the fewest such values; ties keep the order as
// written.  (Deeper-first, left-first and right-first orders were tried too: on generated circuits they win in one
// statement in a hundred, for a third of the host time of this pass.)
// Ops that define nothing (OUTPUT, CHECK) keep their relative order, so the k-th CHECK stays the k-th constraint.
void order_within_statements(std::vector<SOp>& ops, const std::vector<uint32_t>& stmt_begin, int32_t nv) {
  const size_t n = ops.size();
  std::vector<uint32_t> begin;
  for (uint32_t b : stmt_begin)
    if (b < n && (begin.empty() || b > begin.back())) begin.push_back(b);
  if (begin.empty() || begin[0] != 0) begin.insert(begin.begin(), 0u);
  begin.push_back((uint32_t)n);
  auto defines = [](const SOp& o) { return !is_output_op(o.code) && o.code != ZKIR_OP_CHECK && o.code != ZKIR_OP_NOP; };
  auto sources = [](const SOp& o, Ref out[3]) {
    out[0] = out[1] = out[2] = kNoRef;
    if (is_input_op(o.code) || o.code == ZKIR_OP_NOP) return;
    out[0] = o.a;
    out[1] = o.b;
    if (o.code == ZKIR_OP_CHECK) out[2] = o.dst;
  };
  static thread_local struct {
    std::vector<uint32_t> total_uses;
    std::vector<int32_t> local_of;
  } W;  // (memory kept from call to call, per thread)
  std::vector<uint32_t>& total_uses = W.total_uses;
  total_uses.assign(nv, 0);
  for (const SOp& o : ops) {
    Ref src[3];
    sources(o, src);
    for (Ref r : src)
      if (r >= 0) total_uses[r]++;
  }
  constexpr size_t kMaxBlock = 48;  // longer statements (deep expressions) stay as written
  static thread_local std::unordered_map<uint64_t, std::array<uint8_t, kMaxBlock + 1>> shape_cache;
  std::vector<int32_t>& local_of = W.local_of;  // value -> index within the block of the op that defines it
  local_of.assign(nv, -1);
  SOp moved[kMaxBlock];
  for (size_t bi = 0; bi + 1 < begin.size(); bi++) {
    const uint32_t i0 = begin[bi], m = begin[bi + 1] - i0;
    if (m < 4 || m > kMaxBlock) continue;
    const SOp* blk = ops.data() + i0;
    int32_t prod[kMaxBlock][3];   // producers within the block of each op's sources
    uint8_t depth[kMaxBlock];
    uint32_t uses_in[kMaxBlock];
    bool is_root[kMaxBlock];
    for (uint32_t j = 0; j < m; j++)
      if (defines(blk[j]) && blk[j].dst >= 0) local_of[blk[j].dst] = (int32_t)j;
    for (uint32_t j = 0; j < m; j++) {
      uses_in[j] = 0;
      is_root[j] = true;
    }
    bool forward_only = true;
    for (uint32_t j = 0; j < m; j++) {
      Ref src[3];
      sources(blk[j], src);
      depth[j] = 0;
      for (int s = 0; s < 3; s++) {
        const int32_t p = src[s] >= 0 ? local_of[src[s]] : -1;
        prod[j][s] = p;
        if (p < 0) continue;
        if ((uint32_t)p >= j) forward_only = false;  // (cannot happen: a value is defined before it is read)
        else {
          uses_in[p]++;
          is_root[p] = false;
          depth[j] = std::max<uint8_t>(depth[j], (uint8_t)(depth[p] + 1));
        }
      }
    }
    auto cost_of = [&](const uint32_t* order) -> uint32_t {
      // position of every op, then per defined value: is every read of it the accumulator's (no op that defines
      // something between its definition and the read)?  The others are stored from their definition to their last read.
      uint32_t pos[kMaxBlock], n_def_before[kMaxBlock + 1];
      n_def_before[0] = 0;
      for (uint32_t q = 0; q < m; q++) {
        pos[order[q]] = q;
        n_def_before[q + 1] = n_def_before[q] + (defines(blk[order[q]]) ? 1u : 0u);
      }
      uint32_t last_use[kMaxBlock];
      bool stored[kMaxBlock];
      for (uint32_t j = 0; j < m; j++) {
        last_use[j] = pos[j];
        const bool escapes = defines(blk[j]) && blk[j].dst >= 0 && total_uses[blk[j].dst] > uses_in[j];
        stored[j] = escapes;
        if (escapes) last_use[j] = m;
      }
      for (uint32_t j = 0; j < m; j++)
        for (int s = 0; s < 3; s++) {
          const int32_t p = prod[j][s];
          if (p < 0) continue;
          last_use[p] = std::max(last_use[p], pos[j]);
          if (n_def_before[pos[j]] - n_def_before[pos[p] + 1] != 0) stored[p] = true;
        }
      int32_t delta[kMaxBlock + 2] = {0};
      uint32_t n_stored = 0;
      for (uint32_t j = 0; j < m; j++)
        if (stored[j] && defines(blk[j])) {
          n_stored++;
          delta[pos[j]]++;
          delta[last_use[j] + 1]--;
        }
      int32_t live = 0, peak = 0;
      for (uint32_t q = 0; q <= m; q++) {
        live += delta[q];
        peak = std::max(peak, live);
      }
      return (uint32_t)peak * 256u + n_stored;
    };
    // only an op that reads two values made in this statement leaves a choice
    bool any_choice = false;
    for (uint32_t j = 0; j < m && !any_choice; j++) {
      int np = 0;
      for (int s = 0; s < 3; s++) np += prod[j][s] >= 0 && (s == 0 || prod[j][s] != prod[j][0]);
      any_choice = np >= 2;
    }
    if (!any_choice) forward_only = false;
    uint32_t order[kMaxBlock], best_order[kMaxBlock];
    for (uint32_t j = 0; j < m; j++) best_order[j] = j;
    // Generated circuits repeat a handful of statement shapes thousands of times: the answer is remembered per shape
    // (which ops define a value, which op of the statement makes each source, which values are read elsewhere).  A
    // remembered order is checked against this statement before it is used, so two shapes with one key cost a
    // good order, never a wrong tape.
    uint64_t key = 1469598103934665603ull ^ m;
    if (forward_only) {
      for (uint32_t j = 0; j < m; j++) {
        const bool def = defines(blk[j]);
        const uint32_t word = (uint32_t)(prod[j][0] + 1) | (uint32_t)(prod[j][1] + 1) << 6 | (uint32_t)(prod[j][2] + 1) << 12 |
                              (def ? 1u << 18 : 0u) | (def && blk[j].dst >= 0 && total_uses[blk[j].dst] > uses_in[j] ? 1u << 19 : 0u);
        key = (key ^ word) * 1099511628211ull;
      }
      const auto hit = shape_cache.find(key);
      if (hit != shape_cache.end() && hit->second[0] == m) {
        bool ok = true;
        uint32_t pos[kMaxBlock];
        uint64_t seen = 0;
        for (uint32_t q = 0; q < m && ok; q++) {
          const uint32_t j = hit->second[q + 1];
          ok = j < m && !(seen >> j & 1);
          if (ok) {
            seen |= 1ull << j;
            pos[j] = q;
          }
        }
        for (uint32_t j = 0; j < m && ok; j++)
          for (int s = 0; s < 3; s++) ok = ok && (prod[j][s] < 0 || pos[prod[j][s]] < pos[j]);
        if (ok) {
          for (uint32_t q = 0; q < m; q++) best_order[q] = hit->second[q + 1];
          forward_only = false;  // nothing left to search
        }
      }
    }
    const bool searched = forward_only;
    if (forward_only) {
      bool emitted[kMaxBlock] = {false};
      uint32_t n_out = 0;
      // depth-first from the ops nothing in the statement reads, in the order written; explicit stack of (op, next source)
      uint32_t stack_op[kMaxBlock], stack_at[kMaxBlock];
      for (uint32_t root = 0; root < m; root++) {
        if (!is_root[root] || emitted[root]) continue;
        uint32_t sp = 0;
        stack_op[sp] = root;
        stack_at[sp++] = 0;
        while (sp) {
          const uint32_t j = stack_op[sp - 1];
          // the producers of j, the shallowest first (insertion sort of at most three, stable)
          int32_t ps[3];
          int np = 0;
          for (int s = 0; s < 3; s++)
            if (prod[j][s] >= 0) ps[np++] = prod[j][s];
          for (int x = 1; x < np; x++)
            for (int y = x; y > 0 && depth[ps[y]] < depth[ps[y - 1]]; y--) std::swap(ps[y], ps[y - 1]);
          bool pushed = false;
          while ((int)stack_at[sp - 1] < np) {
            const int32_t p = ps[stack_at[sp - 1]++];
            if (!emitted[p]) {
              emitted[p] = true;  // (marks "on the stack or out": an op is pushed once)
              stack_op[sp] = (uint32_t)p;
              stack_at[sp++] = 0;
              pushed = true;
              break;
            }
          }
          if (pushed) continue;
          emitted[j] = true;
          order[n_out++] = j;
          sp--;
        }
      }
      bool same = n_out == m;
      for (uint32_t q = 0; q < m && same; q++) same = order[q] == q;
      if (n_out == m && !same && cost_of(order) < cost_of(best_order)) std::copy(order, order + m, best_order);
    }
    if (searched) {
      if (shape_cache.size() > 4096) shape_cache.clear();
      std::array<uint8_t, kMaxBlock + 1>& entry = shape_cache[key];
      entry[0] = (uint8_t)m;
      for (uint32_t q = 0; q < m; q++) entry[q + 1] = (uint8_t)best_order[q];
    }
    for (uint32_t j = 0; j < m; j++)
      if (defines(blk[j]) && blk[j].dst >= 0) local_of[blk[j].dst] = -1;
    bool changed = false;
    for (uint32_t j = 0; j < m; j++) changed = changed || best_order[j] != j;
    if (!changed) continue;
    // ops that define nothing keep the order they were written in (constraint numbers follow the CHECKs)
    uint32_t sinks[kMaxBlock], n_sinks = 0;
    for (uint32_t j = 0; j < m; j++)
      if (!defines(blk[j])) sinks[n_sinks++] = j;
    uint32_t at = 0;
    bool sinks_in_order = true;
    for (uint32_t q = 0; q < m; q++)
      if (!defines(blk[best_order[q]])) sinks_in_order = sinks_in_order && best_order[q] == sinks[at++];
    if (!sinks_in_order) continue;
    for (uint32_t q = 0; q < m; q++) moved[q] = blk[best_order[q]];
    std::copy(moved, moved + m, ops.begin() + i0);
  }
}

// ---- strands: which op runs in which of the four op streams, and between which barriers ----
// The strands advance in epochs (the stretches between barriers): an op may use a value of
// its own strand at once, a value of another strand only in a later epoch.  The schedule is
// built epoch by epoch with a clock per strand (a product, a check or an input costs 12
// units, an output conversion 6, everything else 1): the strand that is least far along takes
// the next op it may run, in program order; a strand with nothing to run waits.  The epoch
// ends when every strand waits, or when half of them wait while ops exist that only a barrier
// can release (they read values two strands made in this epoch, or sit behind a busy strand).
// Independent chains of a circuit spread over the strands that way, each epoch ends with the
// strands about equally far along, and a barrier falls where values have to cross.
struct Placement {
  std::vector<uint8_t> strand;             // per op
  std::vector<int32_t> epoch;              // per op
  std::vector<uint32_t> order[ZKIR_MAX_STRANDS];  // ops of each strand in the order they run
  int32_t n_epochs = 1;
};

Placement place_strands(const std::vector<SOp>& ops, int32_t nv) {
  const size_t n = ops.size();
  Placement pl;
  pl.strand.assign(n, 0);
  pl.epoch.assign(n, -1);
  auto sources = [&](const SOp& o, Ref out[3]) {
    out[0] = out[1] = out[2] = kNoRef;
    if (is_input_op(o.code) || o.code == ZKIR_OP_NOP) return;
    out[0] = o.a;
    out[1] = o.b;
    if (o.code == ZKIR_OP_CHECK) out[2] = o.dst;
  };
  auto defines = [](const SOp& o) { return !is_output_op(o.code) && o.code != ZKIR_OP_CHECK && o.code != ZKIR_OP_NOP; };
  auto cost = [](uint8_t code) -> uint64_t {
    switch (code) {
      case ZKIR_OP_MUL: case ZKIR_OP_CHECK: case ZKIR_OP_INPUT: return 12;
      case ZKIR_OP_OUTPUT_SCALED: case ZKIR_OP_INPUT_SCALED: return 1;
      case ZKIR_OP_OUTPUT: return 6;
      default: return 1;
    }
  };
  // who reads each op's value (ops are single assignment and in a valid order: definitions first)
  std::vector<int32_t> def_op(nv, -1);
  std::vector<uint32_t> n_wait(n, 0), succ_begin(n + 1, 0);
  for (size_t i = 0; i < n; i++) {
    Ref src[3];
    sources(ops[i], src);
    for (int k = 0; k < 3; k++) {
      if (src[k] < 0 || def_op[src[k]] < 0) continue;
      bool seen = false;  // one edge per distinct producer
      for (int q = 0; q < k; q++) seen = seen || (src[q] >= 0 && def_op[src[q]] == def_op[src[k]]);
      if (seen) continue;
      n_wait[i]++;
      succ_begin[(size_t)def_op[src[k]] + 1]++;
    }
    if (defines(ops[i]) && ops[i].dst >= 0) def_op[ops[i].dst] = (int32_t)i;
  }
  for (size_t i = 0; i < n; i++) succ_begin[i + 1] += succ_begin[i];
  std::vector<uint32_t> succ(succ_begin[n]), fill(succ_begin.begin(), succ_begin.end() - 1);
  std::fill(def_op.begin(), def_op.end(), -1);
  for (size_t i = 0; i < n; i++) {
    Ref src[3];
    sources(ops[i], src);
    for (int k = 0; k < 3; k++) {
      if (src[k] < 0 || def_op[src[k]] < 0) continue;
      bool seen = false;
      for (int q = 0; q < k; q++) seen = seen || (src[q] >= 0 && def_op[src[q]] == def_op[src[k]]);
      if (!seen) succ[fill[def_op[src[k]]]++] = (uint32_t)i;
    }
    if (defines(ops[i]) && ops[i].dst >= 0) def_op[ops[i].dst] = (int32_t)i;
  }

  typedef std::priority_queue<uint32_t, std::vector<uint32_t>, std::greater<uint32_t>> Heap;  // smallest index first
  Heap anywhere, bound[ZKIR_STRANDS];
  std::vector<uint32_t> after_barrier;  // runnable, but its operands come from two strands of this epoch
  // the strand whose value of the current epoch an unplaced op reads: -1 none, -2 more than one
  std::vector<int32_t> tie(n, -1), tie_epoch(n, -1);
  size_t left = 0;
  for (size_t i = 0; i < n; i++) {
    if (ops[i].code == ZKIR_OP_NOP) {
      pl.epoch[i] = 0;
      continue;
    }
    left++;
    if (n_wait[i] == 0) anywhere.push((uint32_t)i);
  }
  uint64_t clock[ZKIR_STRANDS] = {0, 0, 0, 0};
  bool idle[ZKIR_STRANDS] = {false, false, false, false};
  int32_t e = 0;
  size_t placed_in_epoch = 0;
  const uint64_t kBarrier = 2;
  auto close_epoch = [&]() {
    uint64_t t = 0;
    for (uint64_t c : clock) t = std::max(t, c);
    for (int s = 0; s < (int)ZKIR_STRANDS; s++) {
      clock[s] = t + kBarrier;
      idle[s] = false;
      while (!bound[s].empty()) {  // what was tied to a strand by a value of this epoch is free to move now
        anywhere.push(bound[s].top());
        bound[s].pop();
      }
    }
    for (uint32_t i : after_barrier) anywhere.push(i);
    after_barrier.clear();
    e++;
  };
  while (left) {
    int s = -1, n_idle = 0;
    for (int k = 0; k < (int)ZKIR_STRANDS; k++) {
      if (idle[k]) {
        n_idle++;
        continue;
      }
      if (s < 0 || clock[k] < clock[s]) s = k;
    }
    bool waiting = !after_barrier.empty();
    for (int k = 0; k < (int)ZKIR_STRANDS && !waiting; k++) waiting = !bound[k].empty() && idle[k] == false && n_idle > 0;
    if (s < 0 || (n_idle * 2 >= (int)ZKIR_STRANDS && waiting)) {
      // two barriers in a row with nothing placed between them: no op can ever run (cannot happen
      // for single-assignment ops in definition order; better an error than a loop)
      if (placed_in_epoch == 0 && e > 0 && anywhere.empty() && after_barrier.empty() && s < 0)
        throw LowerError(ZKIR_LOWER_BAD_AST, "internal: strand scheduling made no progress");
      close_epoch();
      placed_in_epoch = 0;
      continue;
    }
    uint32_t i;
    if (!bound[s].empty() && (anywhere.empty() || bound[s].top() < anywhere.top())) {
      i = bound[s].top();
      bound[s].pop();
    } else if (!anywhere.empty()) {
      i = anywhere.top();
      anywhere.pop();
    } else {
      idle[s] = true;
      continue;
    }
    pl.strand[i] = (uint8_t)s;
    pl.epoch[i] = e;
    pl.order[s].push_back(i);
    clock[s] += cost(ops[i].code);
    left--;
    placed_in_epoch++;
    for (uint32_t q = succ_begin[i]; q < succ_begin[i + 1]; q++) {
      const uint32_t j = succ[q];
      if (tie_epoch[j] != e) {
        tie_epoch[j] = e;
        tie[j] = -1;
      }
      tie[j] = tie[j] == -1 || tie[j] == s ? s : -2;
      if (--n_wait[j] == 0) {
        if (tie[j] == -2)
          after_barrier.push_back(j);
        else
          bound[tie[j]].push(j);  // its last producer ran in this epoch, so a tie exists
      }
    }
    // a strand that was waiting may have work now only through `anywhere`, which this op cannot
    // have fed (its readers are tied to strand s): nobody wakes up here
  }
  pl.n_epochs = e + 1;

  // Second look: an op that ran next to its operand's producer while its only reader sits in another
  // strand sends its result across.  When several such ops hang on the same value (a value multiplied
  // by different constants for different sums), it is that one value that should cross, and each op
  // should run where its reader is, right before it: fewer exchange slots, the same work per epoch.
  std::vector<uint32_t> n_readers(n, 0), reader(n, 0);
  for (size_t i = 0; i < n; i++)
    for (uint32_t q = succ_begin[i]; q < succ_begin[i + 1]; q++) {
      n_readers[i]++;
      reader[i] = succ[q];
    }
  std::fill(def_op.begin(), def_op.end(), -1);
  for (size_t i = 0; i < n; i++)
    if (defines(ops[i]) && ops[i].dst >= 0) def_op[ops[i].dst] = (int32_t)i;
  auto may_follow = [&](size_t i, uint32_t c) {  // could op i run right before its reader c, in c's strand?
    if (is_input_op(ops[i].code) || !defines(ops[i]) || n_readers[i] != 1) return false;
    if (pl.strand[c] == pl.strand[i]) return false;
    Ref src[3];
    sources(ops[i], src);
    bool has_value = false;
    for (Ref r : src) {
      if (r < 0 || def_op[r] < 0) continue;
      has_value = true;
      if (pl.epoch[def_op[r]] >= pl.epoch[c]) return false;  // its operand must have crossed a barrier by then
    }
    return has_value;
  };
  std::vector<uint32_t> n_followers(n, 0);  // per producer: readers that could follow their own reader
  for (size_t i = 0; i < n; i++) {
    if (ops[i].code == ZKIR_OP_NOP || !may_follow(i, reader[i])) continue;
    Ref src[3];
    sources(ops[i], src);
    for (Ref r : src)
      if (r >= 0 && def_op[r] >= 0) n_followers[def_op[r]]++;
  }
  std::vector<std::vector<uint32_t>> before(n);  // ops that now run right before op c
  std::vector<bool> moved(n, false);
  bool any = false;
  for (size_t k = n; k-- > 0;) {  // readers first: their own place is final by the time their operands are looked at
    const size_t i = k;
    if (ops[i].code == ZKIR_OP_NOP) continue;
    const uint32_t c = reader[i];
    if (!may_follow(i, c)) continue;
    Ref src[3];
    sources(ops[i], src);
    bool shared = true;
    for (Ref r : src)
      if (r >= 0 && def_op[r] >= 0) shared = shared && n_followers[def_op[r]] >= 2;
    if (!shared) continue;
    pl.strand[i] = pl.strand[c];
    pl.epoch[i] = pl.epoch[c];
    before[c].push_back((uint32_t)i);
    moved[i] = true;
    any = true;
  }
  if (any) {
    std::vector<uint32_t> fresh[ZKIR_STRANDS];
    // an op and, ahead of it, what follows it (in program order, which is a valid order among them)
    std::vector<uint32_t> stack;
    for (int s = 0; s < (int)ZKIR_STRANDS; s++) {
      for (uint32_t i : pl.order[s]) {
        if (moved[i]) continue;
        // depth-first: the ops before i, each preceded by its own followers
        std::vector<uint32_t> seq;
        stack.assign(1, i);
        while (!stack.empty()) {
          const uint32_t x = stack.back();
          stack.pop_back();
          seq.push_back(x);
          for (uint32_t y : before[x]) stack.push_back(y);
        }
        std::sort(seq.begin(), seq.end());  // program order: definitions before uses
        fresh[s].insert(fresh[s].end(), seq.begin(), seq.end());
      }
      pl.order[s].swap(fresh[s]);
    }
  }
  return pl;
}

void write_tape(Tape& tape, const Pass1& p1, const std::vector<zkir_op>& out_ops, const std::vector<uint32_t>& triples) {
  const size_t n = out_ops.size();
  tape.blob.clear();
  tape.blob.reserve(n * sizeof(zkir_op) + p1.consts.size() * 32 + triples.size() * 4);  // (appended to, not zero-filled first)
  const uint8_t* ob = (const uint8_t*)out_ops.data();
  tape.blob.insert(tape.blob.end(), ob, ob + n * sizeof(zkir_op));
  for (const U256& c : p1.consts) {
    uint32_t l[8];
    c.limbs32(l);
    tape.blob.insert(tape.blob.end(), (const uint8_t*)l, (const uint8_t*)l + 32);
  }
  const uint8_t* tb = (const uint8_t*)triples.data();
  tape.blob.insert(tape.blob.end(), tb, tb + triples.size() * 4);
  tape.inputs = p1.inputs;
  tape.outputs = p1.outputs;
}


// ---- scaled values ---------------------------------------------------------------------------------------------
// The device holds a tape value x as the words of x * 2^256 mod p (Montgomery form), so every output costs a
// reduction (0.53 of a product) and every input a product.  A value v of the circuit may instead live on the tape as
// v * 2^-256 ("scaled"; domain 1 below, plain is domain 0): its words on the device are v's own, INPUT_SCALED and
// OUTPUT_SCALED move it in and out for nothing, sums of scaled values are scaled, and the product of a scaled and a
// plain value is scaled.  What it costs: a product needs one plain factor, so a signal that is used as the plain
// factor of some product gets a plain copy, made by one product with the constant 2^256 (once: the copy is a
// value like any other).  This pass rewrites the single-assignment ops of a + - * circuit with a domain per value,
// greedily in program order: every op takes the cheapest combination of its operands' domains (a copy that exists
// is free, a constant is free in either domain, an input is free scaled), a value that flows into an output
// through sums prefers to be scaled; of two factors that would both need a plain copy the one with more products
// ahead of it gets it.  The rewrite is kept only if it is cheaper by that count than leaving every value plain (a
// chain of squarings like the gadget's S-boxes gains nothing: both factors of x * x are the same value).
// Returns the number of products added (0: ops left as they were).
uint32_t assign_scales(Pass1& p1) {
  const std::vector<SOp>& ops = p1.ops;
  const size_t n = ops.size();
  const int32_t nv = p1.n_values;
  for (const SOp& o : ops) {
    const uint8_t c = o.code;
    if (c != ZKIR_OP_INPUT && c != ZKIR_OP_OUTPUT && c != ZKIR_OP_COPY && c != ZKIR_OP_NEG && c != ZKIR_OP_ADD && c != ZKIR_OP_SUB &&
        c != ZKIR_OP_MUL && c != ZKIR_OP_CHECK && c != ZKIR_OP_NOP)
      return 0;
  }
  // in products: a copy in the other domain is one product plus, on average, 1.3 more device-slot accesses (the copy
  // is stored and read back), and on launches that spill a slot access costs about what 0.9 products cost (measured:
  // the all-quadratic benchmark circuit lost 12 % with 331 copies for 1 099 free outputs, the circuit with
  // constant factors won 7 % with 225 copies for 1 047)
  static const float kConvert = std::getenv("ZKIR_SCALE_CONVERT_COST") ? (float)std::atof(std::getenv("ZKIR_SCALE_CONVERT_COST")) : 2.2f;
  constexpr float kOutputPlain = 0.53f, kInputPlain = 1.0f;
  U256 scale, scale_inv;  // 2^256 mod p and its inverse
  {
    U256 two128;
    two128.w[2] = 1;
    scale = fr::mul(two128, two128);
    if (!fr::div(U256::from_u64(1), scale, &scale_inv)) return 0;
  }
  // which values flow into an output through sums, and how often each value is a factor
  std::vector<uint8_t> to_output(nv, 0);
  std::vector<uint32_t> factor_uses(nv, 0);
  std::vector<int32_t> input_of(nv, -1);
  uint32_t n_outputs = 0, n_inputs = 0;
  for (size_t i = n; i-- > 0;) {
    const SOp& o = ops[i];
    switch (o.code) {
      case ZKIR_OP_OUTPUT:
        n_outputs++;
        if (o.a >= 0) to_output[o.a] = 1;
        break;
      case ZKIR_OP_ADD: case ZKIR_OP_SUB: case ZKIR_OP_NEG: case ZKIR_OP_COPY:
        if (o.dst >= 0 && to_output[o.dst]) {
          if (o.a >= 0) to_output[o.a] = 1;
          if (o.b >= 0) to_output[o.b] = 1;
        }
        break;
      case ZKIR_OP_MUL: case ZKIR_OP_CHECK:
        if (o.a >= 0) factor_uses[o.a]++;
        if (o.b >= 0) factor_uses[o.b]++;
        break;
      case ZKIR_OP_INPUT:
        n_inputs++;
        input_of[o.dst] = o.a;
        break;
      default: break;
    }
  }
  struct Rep {
    Ref r[2] = {kNoRef, kNoRef};
  };
  std::vector<Rep> rep(nv);
  std::vector<SOp> out;
  out.reserve(n + n / 4);
  std::vector<uint32_t> new_begin;
  new_begin.reserve(p1.stmt_begin.size());
  int32_t n_new = 0;
  uint32_t n_scale = 0, n_out_plain = 0, n_in_plain = 0;
  std::unordered_map<uint32_t, Ref> scaled_const;  // pool index -> pool entry of the constant times 2^-256

  auto price = [&](Ref r, int e) -> float {
    if (r < 0) return 0.f;  // a constant (or no operand)
    if (rep[r].r[e] != kNoRef) return 0.f;
    if (input_of[r] >= 0) return e ? 0.f : kInputPlain;
    return kConvert;
  };
  auto get = [&](Ref r, int e) -> Ref {
    if (r == kNoRef) return kNoRef;
    if (is_const(r)) {
      if (!e) return r;
      auto it = scaled_const.find(const_index(r));
      if (it != scaled_const.end()) return it->second;
      const Ref s = p1.constant(fr::mul(p1.consts[const_index(r)], scale_inv));
      scaled_const.emplace(const_index(r), s);
      return s;
    }
    if (rep[r].r[e] != kNoRef) return rep[r].r[e];
    const Ref v = n_new++;
    if (input_of[r] >= 0) {
      out.push_back({(uint8_t)(e ? ZKIR_OP_INPUT_SCALED : ZKIR_OP_INPUT), v, (Ref)input_of[r], kNoRef});
      n_in_plain += e ? 0 : 1;
    } else {
      // the other copy times 2^256 (scaled -> plain) or 2^-256 (plain -> scaled)
      out.push_back({ZKIR_OP_MUL, v, rep[r].r[1 - e], p1.constant(e ? scale_inv : scale)});
      n_scale++;
    }
    rep[r].r[e] = v;
    return v;
  };
  auto define = [&](Ref old, int e) -> Ref {
    const Ref v = n_new++;
    rep[old].r[e] = v;
    return v;
  };
  size_t next_begin = 0;
  for (size_t i = 0; i < n; i++) {
    while (next_begin < p1.stmt_begin.size() && p1.stmt_begin[next_begin] <= i) {
      if (p1.stmt_begin[next_begin] == i && (new_begin.empty() || new_begin.back() != out.size())) new_begin.push_back((uint32_t)out.size());
      next_begin++;
    }
    const SOp& o = ops[i];
    switch (o.code) {
      case ZKIR_OP_NOP: case ZKIR_OP_INPUT:  // (an input is loaded where it is first read, in the domain it is read in)
        break;
      case ZKIR_OP_OUTPUT: {
        const int e = price(o.a, 1) <= price(o.a, 0) + kOutputPlain ? 1 : 0;
        const Ref a = get(o.a, e);
        out.push_back({(uint8_t)(e ? ZKIR_OP_OUTPUT_SCALED : ZKIR_OP_OUTPUT), o.dst, a, kNoRef});
        n_out_plain += e ? 0 : 1;
        break;
      }
      case ZKIR_OP_ADD: case ZKIR_OP_SUB: case ZKIR_OP_NEG: case ZKIR_OP_COPY: {
        float c[2];
        for (int e = 0; e < 2; e++) c[e] = price(o.a, e) + price(o.b, e) + (!e && to_output[o.dst] ? kOutputPlain : 0.f);
        const int e = c[1] < c[0] ? 1 : 0;
        const Ref a = get(o.a, e), b = get(o.b, e);
        out.push_back({o.code, define(o.dst, e), a, b});
        break;
      }
      case ZKIR_OP_MUL: case ZKIR_OP_CHECK: {
        // (domain of a, domain of b); the product's domain is their sum
        static const int kA[3] = {0, 1, 0}, kB[3] = {0, 0, 1};
        float best = 1e30f;
        int pick = 0;
        for (int k = 0; k < 3; k++) {
          const int ea = kA[k], eb = kB[k], e = ea + eb;
          float c = price(o.a, ea) + (o.b == o.a && ea == eb ? 0.f : price(o.b, eb));
          if (o.code == ZKIR_OP_CHECK) c += price(o.dst, e);
          else if (!e && to_output[o.dst]) c += kOutputPlain;
          // of two factors that would both need a plain copy, the one with more products ahead gets it
          if (k == 1 && o.b >= 0 && rep[o.b].r[0] == kNoRef) c -= 1e-4f * (float)std::min<uint32_t>(factor_uses[o.b], 1000u);
          if (k == 2 && o.a >= 0 && rep[o.a].r[0] == kNoRef) c -= 1e-4f * (float)std::min<uint32_t>(factor_uses[o.a], 1000u);
          if (c < best) {
            best = c;
            pick = k;
          }
        }
        const int ea = kA[pick], eb = kB[pick];
        const Ref a = get(o.a, ea), b = get(o.b, eb);
        if (o.a >= 0) factor_uses[o.a]--;
        if (o.b >= 0) factor_uses[o.b]--;
        if (o.code == ZKIR_OP_CHECK) out.push_back({ZKIR_OP_CHECK, get(o.dst, ea + eb), a, b});
        else out.push_back({ZKIR_OP_MUL, define(o.dst, ea + eb), a, b});
        break;
      }
      default: return 0;
    }
  }
  const float cost_scaled = kConvert * n_scale + kOutputPlain * n_out_plain + kInputPlain * n_in_plain;
  const float cost_plain = kOutputPlain * n_outputs + kInputPlain * n_inputs;
  if (std::getenv("ZKIR_SCALE_DEBUG"))
    fprintf(stderr, "assign_scales: %u copies, %u of %u outputs plain, %u of %u inputs plain: cost %.1f against %.1f\n", n_scale,
            n_out_plain, n_outputs, n_in_plain, n_inputs, cost_scaled, cost_plain);
  if (!(cost_scaled < 0.97f * cost_plain)) return 0;
  new_begin.push_back((uint32_t)out.size());
  p1.ops.swap(out);
  p1.stmt_begin.swap(new_begin);
  p1.n_values = n_new;
  return n_scale;
}

void check_options(const LowerOptions& opts_) {
  if (opts_.n_lslots > ZKIR_MAX_LSLOTS) throw LowerError(ZKIR_LOWER_BAD_AST, "n_lslots above ZKIR_MAX_LSLOTS");
  if (opts_.strands != 0 && opts_.strands != ZKIR_STRANDS && opts_.strands != ZKIR_MAX_STRANDS)
    throw LowerError(ZKIR_LOWER_BAD_AST, "strands must be 0, 4 or 8");
}

// pass 2 and the tape itself, from the single-assignment ops of pass 1
Tape place_tape(Pass1& p1, const LowerOptions& opts_) {
  const uint32_t n_scale_ops = opts_.scaled_values ? assign_scales(p1) : 0;
  if (opts_.order_ops) order_within_statements(p1.ops, p1.stmt_begin, p1.n_values);
  const std::vector<SOp>& ops = p1.ops;
  const size_t n = ops.size();
  const int32_t nv = p1.n_values;

  // constraint k is the k-th CHECK in program order, whatever stream it ends up in
  std::vector<uint32_t> check_id(n, 0);
  uint32_t n_checks = 0;
  bool plain_ops_only = true;  // + - * and data movement: what the strand kernel runs
  for (size_t i = 0; i < n; i++) {
    if (ops[i].code == ZKIR_OP_CHECK) check_id[i] = n_checks++;
    const uint8_t c = ops[i].code;
    if (c == ZKIR_OP_BNOT || c == ZKIR_OP_BNOT256 || (c >= ZKIR_OP_POW && c <= ZKIR_OP_SHL)) plain_ops_only = false;
  }

  Tape tape;
  zkir_tape_hdr& h = tape.hdr;
  h.magic = ZKIR_TAPE_MAGIC;
  h.version = ZKIR_TAPE_VERSION;
  h.n_inputs = (uint32_t)p1.n_inputs;
  h.n_outputs = (uint32_t)p1.n_outputs;
  h.n_consts = (uint32_t)p1.consts.size();
  h.n_constraints = n_checks;
  h.n_scale_ops = n_scale_ops;
  std::vector<uint32_t> triples((size_t)n_checks * 3, 0);

  if (opts_.strands && plain_ops_only && n_checks < (1u << 24) - 1) {
    const int S = (int)opts_.strands;
    const uint32_t max_x = std::min<uint32_t>(opts_.max_xslots ? opts_.max_xslots : (S == (int)ZKIR_MAX_STRANDS ? ZKIR_MAX_XSLOTS : 32u), ZKIR_MAX_XSLOTS);
    Placement pl = place_strands(ops, nv);
    if (S == (int)ZKIR_MAX_STRANDS) {
      // Eight strands: the four of the schedule above keep every op that defines a value; the ops that define
      // nothing (OUTPUT, CHECK: more than a third of the work of a circuit of rank-1 constraints) move from strand s
      // to strand s + 4, one epoch later, when everything they read has crossed a barrier.  The chains of the
      // circuit advance on four wavefronts while four more check and convert what the last epoch produced: same
      // barriers, and the two wavefronts that share a SIMD fill each other's issue gaps.
      bool any_in_last = false;
      for (int s = 0; s < (int)ZKIR_STRANDS; s++) {
        std::vector<uint32_t> keep;
        for (uint32_t i : pl.order[s]) {
          if (is_output_op(ops[i].code) || ops[i].code == ZKIR_OP_CHECK) {
            any_in_last = any_in_last || pl.epoch[i] == pl.n_epochs - 1;
            pl.strand[i] = (uint8_t)(s + (int)ZKIR_STRANDS);
            pl.epoch[i] += 1;
            pl.order[s + ZKIR_STRANDS].push_back(i);
          } else {
            keep.push_back(i);
          }
        }
        pl.order[s].swap(keep);
      }
      if (any_in_last) pl.n_epochs += 1;
    }
    // values read outside the strand that defines them live in exchange slots.  A slot is free
    // again for a definition in an epoch after the one of its value's last read: every reader
    // has then passed the barrier the writer waits at.
    std::vector<int32_t> def_op(nv, -1), last_read_epoch(nv, -1);
    std::vector<bool> crosses(nv, false);
    auto note = [&](Ref r, size_t at) {
      if (r < 0 || def_op[r] < 0) return;
      last_read_epoch[r] = std::max(last_read_epoch[r], pl.epoch[at]);
      if (pl.strand[def_op[r]] != pl.strand[at]) crosses[r] = true;
    };
    for (size_t i = 0; i < n; i++) {
      const SOp& o = ops[i];
      if (!is_input_op(o.code)) {
        note(o.a, i);
        note(o.b, i);
        if (o.code == ZKIR_OP_CHECK) note(o.dst, i);
      }
      const bool defines = !is_output_op(o.code) && o.code != ZKIR_OP_CHECK && o.code != ZKIR_OP_NOP;
      if (defines && o.dst >= 0) def_op[o.dst] = (int32_t)i;
    }
    std::vector<uint32_t> xhome(nv, ZKIR_OPERAND_NONE);
    std::vector<int32_t> slot_free_from;  // per exchange slot: first epoch a new value may be defined in
    bool fits = true;
    for (size_t i = 0; i < n && fits; i++) {
      const SOp& o = ops[i];
      const bool defines = !is_output_op(o.code) && o.code != ZKIR_OP_CHECK && o.code != ZKIR_OP_NOP;
      if (!defines || o.dst < 0 || !crosses[o.dst]) continue;
      uint32_t slot = UINT32_MAX;
      for (uint32_t s = 0; s < slot_free_from.size(); s++)
        if (slot_free_from[s] <= pl.epoch[i]) {
          slot = s;
          break;
        }
      if (slot == UINT32_MAX) {
        if (slot_free_from.size() >= max_x) {
          fits = false;
          break;
        }
        slot = (uint32_t)slot_free_from.size();
        slot_free_from.push_back(0);
      }
      slot_free_from[slot] = last_read_epoch[o.dst] + 1;
      xhome[o.dst] = ZKIR_OPERAND(ZKIR_KIND_XSLOT, slot);
    }
    if (fits) {
      // each strand's ops in program order, a barrier wherever its epoch advances, and as many
      // barriers at its end as it takes for every strand to hold the same number
      std::vector<SOp> all = ops;
      const uint32_t barrier_at = (uint32_t)all.size();
      all.push_back({ZKIR_OP_BARRIER, kNoRef, kNoRef, kNoRef});
      check_id.push_back(0);
      std::vector<uint32_t> stream[ZKIR_MAX_STRANDS];
      int32_t at_epoch[ZKIR_MAX_STRANDS] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int s = 0; s < S; s++)
        for (uint32_t i : pl.order[s]) {
          for (; at_epoch[s] < pl.epoch[i]; at_epoch[s]++) stream[s].push_back(barrier_at);
          stream[s].push_back(i);
        }
      std::vector<zkir_op> out_ops;
      uint32_t n_mul = 0;
      // A strand tape is for launches short of wavefronts, so its workgroups should fit a CU five at a
      // time (what the kernel's registers allow): 32 KiB of LDS each.  The exchange slots (2 KiB) come
      // first; the private window rows (8 KiB: one row holds a slot of all four strands) get what is
      // left, and the other private values go to device memory, whose latency the additional
      // wavefronts hide.  Measured on the gadget (8 exchange slots): two window rows beat three from
      // 2^17 vectors on (178 against 211 ms) and cost 2 % at 2^16.
      LowerOptions sopts = opts_;
      // Eight strands keep two sets of rows (a row holds four wavefronts), need more exchange slots (what the
      // second four read has to cross) and are for the smallest batches, where a CU holds one workgroup: 96 KiB.
      const uint32_t x_bytes = (uint32_t)slot_free_from.size() * 2048u, budget = (S == 8 ? 96u : 32u) * 1024u;
      const uint32_t row_bytes = 8192u * (S == 8 ? 2u : 1u);
      sopts.n_lslots = std::min<uint32_t>(opts_.n_lslots, x_bytes >= budget ? 0u : (budget - x_bytes) / row_bytes);
      // (a launch sizes its LDS for the most window rows and the most exchange slots among its tapes: two rows at most
      // with eight strands, so that no mix of such tapes exceeds a CU's 160 KiB even in the 29-bit kernel's 9 KiB rows)
      if (S == (int)ZKIR_MAX_STRANDS) sopts.n_lslots = std::min<uint32_t>(sopts.n_lslots, 2u);
      for (int s = 0; s < S; s++) {
        for (; at_epoch[s] < pl.n_epochs - 1; at_epoch[s]++) stream[s].push_back(barrier_at);
        StreamOut so = place_stream(all, stream[s], nv, xhome, check_id, true, sopts, &triples);
        h.strand_ops[s] = (uint32_t)so.ops.size();
        h.n_gslots = std::max(h.n_gslots, so.n_g);
        h.n_lslots = std::max(h.n_lslots, so.used_l);
        n_mul += so.n_mul;
        out_ops.insert(out_ops.end(), so.ops.begin(), so.ops.end());
      }
      h.flags = ZKIR_TAPE_F_STRANDS;
      h.n_strands = (uint32_t)S;
      h.n_xslots = (uint32_t)slot_free_from.size();
      h.n_ops = (uint32_t)out_ops.size();
      h.n_mul_ops = n_mul;
      write_tape(tape, p1, out_ops, triples);
      return tape;
    }
  }

  std::vector<uint32_t> stream(n);
  for (size_t i = 0; i < n; i++) stream[i] = (uint32_t)i;
  // Program order, and (reorder) two reorderings: statements running ahead by at most 16 statements, and by any
  // number.  Which order moves the fewest values through device memory depends on the circuit (a chain is best left
  // alone, a circuit that reads its signals at random gains most from the widest choice): the kernel's operand
  // cache is modelled on each tape and the cheapest is kept.
  const bool reorder = opts_.reorder && n_checks < (1u << 24) - 1 && p1.stmt_begin.size() > 2;
  StreamOut so;
  uint64_t best_cost = UINT64_MAX;
  std::vector<uint32_t> best_triples;
  uint32_t one_word = UINT32_MAX;  // the pool's constant 1, as the runtime finds it at upload
  for (size_t c = 0; c < p1.consts.size() && one_word == UINT32_MAX; c++)
    if (p1.consts[c] == U256::from_u64(1)) one_word = ZKIR_OPERAND(ZKIR_KIND_CONST, c);
  const size_t lookaheads[3] = {0, 16, SIZE_MAX};
  for (int k = 0; k < (reorder ? 3 : 1); k++) {
    if (k) stream = schedule_statements(ops, p1.stmt_begin, nv, lookaheads[k]);
    StreamOut cand = opts_.window_cache ? place_stream_cached(ops, stream, nv, check_id, reorder, opts_, &triples)
                                        : place_stream(ops, stream, nv, {}, check_id, reorder, opts_, &triples);
    // (with one candidate there is nothing to compare: its traffic is not modelled)
    const uint64_t cost = !reorder ? 0 : device_slot_traffic(cand.ops, one_word) * 4096ull + std::min<uint32_t>(cand.n_g, 4095u);
    if (cost < best_cost) {
      best_cost = cost;
      so = std::move(cand);
      best_triples = triples;
    }
  }
  triples.swap(best_triples);
  h.n_ops = (uint32_t)so.ops.size();
  h.n_gslots = so.n_g;
  h.n_lslots = so.used_l;
  h.n_mul_ops = so.n_mul;
  write_tape(tape, p1, so.ops, triples);
  return tape;
}

}  // namespace


// ---- generate + lower in one go, for campaigns --------------------------------------------------------------
// A campaign makes a circuit only to lower it.  For the configurations campaigns draw (no loops, no If, expressions
// of depth 1, quadratic-aware constraints, + - * only) the generator below makes the same random draws in the same
// order as Generator (generator.cpp follows src/generator.rs:142-601 function by function; so does this, with the
// functions named where they are restated) and hands every statement to pass 1 as soon as it exists: no Instruction
// tree, no Variable objects, no names (their draws are made and dropped), one small arena of expression nodes that
// is reused from statement to statement.  The tape is bit for bit the one Lowerer(Generator(seed, cfg).generate(),
// opts).lower() gives (tests/test_host.py compares them over seeds and configurations), a circuit that lower
// refuses is refused with the same class.
namespace {

class StraightLine {
 public:
  StraightLine(uint64_t seed, const GeneratorConfig& cfg, const LowerOptions& o) : p1(o), rng_(Rng::from_seed(seed)), cfg_(cfg) {
    p1.stream_ids_are_positions();
  }
  Pass1 p1;

  // Generator::generate (generator.rs:142-164)
  void run() {
    const size_t n = rng_.rand(cfg_.min_instructions, cfg_.max_instructions);
    const size_t expect = n + cfg_.forced_inputs + 16;
    p1.reserve_for(expect);
    is_array_.reserve(expect);
    len_.reserve(expect);
    role_.reserve(expect);
    mask_.reserve(expect);
    for (uint32_t k = 0; k < cfg_.forced_inputs; k++) {  // extension: scalar input signals declared before anything else
      skip_name();
      const size_t pos = add_var(false, 0, kInput);
      p1.stream_declare(pos, nullptr);
    }
    for (size_t i = 0; i < n; i++)
      if (generate_instr()) n_instr_ = 0;
    p1.finish();
  }

 private:
  enum Role : uint8_t { kLocal, kInput, kOutput };

  void skip_name() { rng_.skip(cfg_.name_len); }  // rand_string: one draw per character (rng.rs:36-45), the name is not kept

  // ScopeStack::add_var + refresh (generator.cpp), for the one scope a straight-line circuit has: position == id
  size_t add_var(bool is_array, size_t len, Role role) {
    const size_t pos = is_array_.size();
    const bool readable = cfg_.inputs_readable && role == kInput;
    is_array_.push_back(is_array);
    len_.push_back((uint32_t)len);
    role_.push_back(role);
    mask_.push_back(0);
    ready_fields_.resize(pos + 1);
    if (role == kLocal) {
      n_local_++;
      (is_array ? local_arrays_ : local_fields_).push_back((uint32_t)pos);
    }
    if (readable) {
      if (is_array) {  // every element readable from the start: the array counts as written
        mask_[pos] = len >= 64 ? ~0ull : ((1ull << len) - 1);
        touched_list_.push_back((uint32_t)pos);  // positions only grow: the list stays ascending
      } else {
        ready_fields_.set(pos, true);
        ready_signals_.push_back((uint32_t)pos);
      }
    }
    const VariableRole r = role == kLocal ? VariableRole::Local() : VariableRole::Signal(role == kInput ? SignalType::Input : SignalType::Output);
    p1.add_variable(pos, r, is_array, len, nullptr);
    return pos;
  }

  // Generator::generate_instr (generator.rs:166-208): the candidates a straight-line, quadratic-aware configuration has
  bool generate_instr() {
    if (n_instr_ >= cfg_.statement_budget) return false;
    n_instr_ += 1;
    enum Cand { kVarDecl, kAssign, kArrayAssign, kConstraint };
    Cand funcs[4];
    size_t n = 0;
    funcs[n++] = kVarDecl;
    if (n_local_ > 0) funcs[n++] = kAssign;
    if (!local_arrays_.empty()) funcs[n++] = kArrayAssign;
    funcs[n++] = kConstraint;
    arena_.reset();
    switch (funcs[rng_.rand(0, n - 1)]) {
      case kVarDecl: return gen_var_decl_default();
      case kAssign: return gen_assign();
      case kArrayAssign: return gen_array_assign();
      case kConstraint: return gen_constraint_quadratic();
    }
    return false;
  }

  // Generator::gen_var_decl with nothing given (generator.rs:214-267)
  bool gen_var_decl_default() {
    bool is_array = false;
    size_t len = 0;
    if (rng_.rand(0, 1) != 0) {
      is_array = true;
      len = rng_.rand(1, cfg_.max_array_len);
    }
    const Expr* def = nullptr;
    if (!is_array && rng_.rand(1, 10) != 1) def = generate_expr();
    Role role = kLocal;
    if (!def && rng_.rand(0, 1) == 0) {
      role = rng_.rand(0, 1) == 0 ? kInput : kOutput;
      if (role == kOutput) role = kLocal;                        // quadratic-aware: every constraint declares its own output
      if (cfg_.forced_inputs && role == kInput) role = kLocal;   // a fixed set of up-front inputs
    }
    skip_name();
    const size_t pos = add_var(is_array, len, role);
    p1.stream_declare(pos, def);
    return true;
  }

  // generator.rs:269-289
  bool gen_assign() {
    if (local_fields_.empty()) return false;
    const size_t pos = local_fields_[rng_.rand(0, local_fields_.size() - 1)];
    const Expr* e = generate_expr();
    ready_fields_.set(pos, true);  // mark_initialized
    p1.stream_assign(pos, *e);
    return true;
  }

  // generator.rs:291-327
  bool gen_array_assign() {
    const size_t pos = local_arrays_[rng_.rand(0, local_arrays_.size() - 1)];
    const size_t n = len_[pos];
    const Expr* e = generate_expr();
    const size_t idx = rng_.rand(0, n - 1);
    if (mask_[pos] == 0)  // first write to this array: the list of written arrays stays ascending
      touched_list_.insert(std::upper_bound(touched_list_.begin(), touched_list_.end(), (uint32_t)pos), (uint32_t)pos);
    mask_[pos] |= 1ull << idx;
    const Expr index = Expr::Constant((uint64_t)idx);
    p1.stream_array_assign(pos, index, *e);
    return true;
  }

  // Generator::generate_expr at depth 0 with the depth limit 1 (generator.rs:484-543): a constant, an operator over
  // two constants, a variable, an array element
  const Expr* generate_expr() {
    const bool has_field_vars = ready_fields_.count() > 0;
    const bool has_array_vars = !touched_list_.empty();
    static const ExprKind kArith[] = {ExprKind::Constant, ExprKind::Add, ExprKind::Sub, ExprKind::Mul};
    const size_t n_fixed = 4;
    const size_t idx = rng_.rand(0, n_fixed + (has_field_vars ? 1 : 0) + (has_array_vars ? 1 : 0) - 1);
    if (idx == 0) return arena_.make(Expr::Constant(rng_.next()));
    if (idx < n_fixed) {
      const Expr l = Expr::Constant(rng_.next());
      const Expr r = Expr::Constant(rng_.next());
      return arena_.make(Expr::Binary(arena_, kArith[idx], l, r));
    }
    if (has_field_vars && idx == n_fixed) {  // gen_var_ref (generator.rs:545-566)
      VarRef v;
      v.id = ready_fields_.select(rng_.rand(0, ready_fields_.count() - 1));
      return arena_.make(Expr::Var(v));
    }
    return gen_array_index();
  }

  // generator.rs:572-601: one draw per written array, one more to choose among them
  const Expr* gen_array_index() {
    const size_t n = touched_list_.size();
    Rng before = rng_;
    rng_.skip(n);
    const size_t pick = rng_.rand(0, n - 1);
    const size_t pos = touched_list_[pick];
    const size_t written = (size_t)__builtin_popcountll(mask_[pos]);
    size_t idx = before.peek(pick) % ((uint64_t)written + 1);
    if (cfg_.safe_array_index) {  // extension: always a written element
      size_t want = idx % written;
      uint64_t m = mask_[pos];
      while (want--) m &= m - 1;
      idx = (size_t)__builtin_ctzll(m);
    }
    VarRef r;
    r.id = pos;
    r.idx = idx;
    return arena_.make(Expr::Var(r));
  }

  // Generator::gen_linear (extension)
  Expr gen_linear(bool factor) {
    const std::vector<uint32_t>& sigs = ready_signals_;
    const bool always_signal = factor && cfg_.signal_factors;
    auto term = [&]() -> Expr {
      if (!sigs.empty() && (always_signal || rng_.rand(0, 3) != 0)) {
        size_t k = rng_.rand(0, sigs.size() - 1);
        const size_t keep = (size_t)cfg_.forced_inputs + cfg_.signal_window;
        if (cfg_.signal_window && sigs.size() > keep && k >= cfg_.forced_inputs && k < sigs.size() - cfg_.signal_window)
          k = (k % keep < cfg_.forced_inputs) ? k % keep : sigs.size() - keep + k % keep;
        VarRef v;
        v.id = sigs[k];
        return Expr::Var(v);
      }
      return Expr::Constant(rng_.next());
    };
    const Expr t = term();
    if (cfg_.max_linear_terms == 1) return t;
    const uint64_t shape = rng_.rand(0, 2);
    if (shape == 0) return t;
    const Expr u = term();
    return Expr::Binary(arena_, shape == 1 ? ExprKind::Add : ExprKind::Sub, t, u);
  }

  // Generator::gen_constraint_quadratic (extension): `fresh output <== lin * lin + lin`
  bool gen_constraint_quadratic() {
    const Expr a = gen_linear(true);
    const Expr b = gen_linear(true);
    const Expr c = gen_linear(false);
    skip_name();
    const size_t pos = add_var(false, 0, kOutput);
    // mark_initialized: an output field is readable from here on, by expressions and by linear parts
    ready_fields_.set(pos, true);
    if (!ready_signals_.empty() && ready_signals_.back() >= pos) throw LowerError(ZKIR_LOWER_BAD_AST, "internal: signal order");
    ready_signals_.push_back((uint32_t)pos);
    const Expr prod = Expr::Binary(arena_, ExprKind::Mul, a, b);
    const Expr rhs = Expr::Binary(arena_, ExprKind::Add, prod, c);
    p1.stream_declare(pos, nullptr);
    p1.stream_constrain(pos, rhs);
    return true;
  }

  Rng rng_;
  GeneratorConfig cfg_;
  Arena arena_;  // the expression nodes of the statement in the making
  size_t n_instr_ = 0;
  std::vector<uint8_t> is_array_, role_;
  std::vector<uint32_t> len_;
  std::vector<uint64_t> mask_;           // arrays: which elements are written (at most 64 elements)
  std::vector<uint32_t> local_fields_, local_arrays_;  // positions, ascending
  size_t n_local_ = 0;
  OrderedSet ready_fields_;              // Field and initialised (gen_var_ref)
  std::vector<uint32_t> touched_list_;   // arrays with a written element, ascending (gen_array_index)
  std::vector<uint32_t> ready_signals_;  // signal, Field, initialised, ascending: signals become readable in declaration order (gen_linear)
};

}  // namespace

bool can_lower_generated(const GeneratorConfig& cfg, const LowerOptions& opts) {
  return cfg.no_loops && !cfg.enable_if && cfg.max_expr_depth == 1 && cfg.quadratic_aware && cfg.arithmetic_only &&
         cfg.max_array_len <= 64 && opts.fold_constants;
}

Tape lower_generated(uint64_t seed, const GeneratorConfig& cfg, const LowerOptions& opts, double* pass1_seconds) {
  if (!can_lower_generated(cfg, opts)) throw std::invalid_argument("configuration needs the circuit built first");
  validate_generator_config(cfg);
  check_options(opts);
  const auto t0 = std::chrono::steady_clock::now();
  StraightLine g(seed, cfg, opts);
  g.run();
  if (pass1_seconds) *pass1_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return place_tape(g.p1, opts);
}

Tape Lowerer::lower() {
  check_options(opts_);
  Pass1 p1(opts_);
  p1.run(circuit);
  return place_tape(p1, opts_);
}

}  // namespace zkir
