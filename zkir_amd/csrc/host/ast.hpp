// The zkIR data model, one to one with the reference's src/ast.rs: the same
// type names, variants and fields, so code written against the Rust crate reads
// the same here.  Every backend (the Circom printer, the lowerer) consumes these.
//
// Layout.  A campaign generates and lowers thousands of circuits per second and
// per core, so the tree is built for that: expression nodes are plain data in a
// bump arena the Circuit owns (no allocation and no destructor per node, children
// are `const Expr*`), a constant that fits 64 bits (every constant the generator
// draws, src/generator.rs:568-570) is carried as the number and printed on demand,
// the per-element `initialized` flags of a Variable sit inline up to 64 elements,
// and the parts only control-flow statements have hang off one pointer.  What the
// Rust holds by value in a Box / Option / Vec is reached here through a pointer
// that reads the same (`*instr.value`, `if (instr.default_value)`).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <new>
#include <optional>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace zkir {

// Bump allocator for the plain-data nodes of one Circuit.  Nothing is freed before the
// arena goes; copies of a Circuit share it (the nodes are immutable once built).
class Arena {
 public:
  Arena() = default;
  Arena(const Arena&) = delete;
  Arena& operator=(const Arena&) = delete;
  ~Arena() {
    for (char* c : chunks_) ::operator delete(c);
  }
  void* alloc(size_t n, size_t align = 8) {
    size_t pad = (size_t)(-(intptr_t)cur_) & (align - 1);
    if (pad + n > left_) {
      grow(n + align);
      pad = (size_t)(-(intptr_t)cur_) & (align - 1);
    }
    char* p = cur_ + pad;
    cur_ = p + n;
    left_ -= pad + n;
    return p;
  }
  template <class T>
  const T* make(const T& v) {
    static_assert(std::is_trivially_destructible<T>::value, "arena nodes are never destroyed");
    return new (alloc(sizeof(T), alignof(T))) T(v);
  }
  // Forget every node (their memory is reused from the first chunk on): for a caller that builds one statement at
  // a time and is done with its nodes before the next.
  void reset() {
    while (chunks_.size() > 1) {
      ::operator delete(chunks_.back());
      chunks_.pop_back();
    }
    cur_ = chunks_.empty() ? nullptr : chunks_[0];
    left_ = chunks_.empty() ? 0 : first_size_;
  }
  const char* text(const char* s, size_t n) {
    char* p = (char*)alloc(n + 1, 1);
    std::memcpy(p, s, n);
    p[n] = 0;
    return p;
  }

 private:
  void grow(size_t at_least) {
    size_t n = next_;
    if (next_ < (1u << 20)) next_ *= 2;
    if (n < at_least) n = at_least;
    chunks_.push_back((char*)::operator new(n));
    if (chunks_.size() == 1) first_size_ = n;
    cur_ = chunks_.back();
    left_ = n;
  }
  std::vector<char*> chunks_;
  char* cur_ = nullptr;
  size_t left_ = 0, next_ = 16384, first_size_ = 0;
};

// src/ast.rs:3-10
struct VarRef {
  size_t id = 0;                 // unique id of the variable
  std::optional<size_t> idx;     // element of an array; arrays are never referenced whole
};

// src/ast.rs:12: a field element is carried as its decimal text.  Text that is the plain decimal form of a
// 64-bit number is held as that number (str() gives the same characters back); anything else (longer, signed,
// leading zeros, not a number at all) is kept verbatim in the circuit's arena.
class Field {
 public:
  Field() = default;
  static Field from_u64(uint64_t v) {
    Field f;
    f.v_ = v;
    return f;
  }
  static Field from_text(Arena& a, const std::string& s) {
    Field f;
    if (plain_u64(s, &f.v_)) return f;
    f.text_ = a.text(s.data(), s.size());
    f.len_ = (uint32_t)s.size();
    return f;
  }
  bool is_u64() const { return text_ == nullptr; }
  uint64_t u64() const { return v_; }
  std::string str() const { return text_ ? std::string(text_, len_) : std::to_string(v_); }
  bool operator==(const Field& o) const { return str() == o.str(); }

 private:
  static bool plain_u64(const std::string& s, uint64_t* out) {
    if (s.empty() || s.size() > 20 || (s.size() > 1 && s[0] == '0')) return false;
    uint64_t v = 0;
    for (char c : s) {
      if (c < '0' || c > '9') return false;
      const uint64_t d = (uint64_t)(c - '0');
      if (v > (UINT64_MAX - d) / 10) return false;
      v = v * 10 + d;
    }
    *out = v;
    return true;
  }
  uint64_t v_ = 0;
  const char* text_ = nullptr;
  uint32_t len_ = 0;
};

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
  const Expr* left = nullptr;     // binary ops, and the operand of BitNot (a Box in the Rust: here a node of the arena)
  const Expr* right = nullptr;    // binary ops

  static Expr Var(VarRef v) { Expr e; e.kind = ExprKind::Var; e.var = v; return e; }
  static Expr Constant(Field f) { Expr e; e.kind = ExprKind::Constant; e.constant = f; return e; }
  static Expr Constant(uint64_t v) { return Constant(Field::from_u64(v)); }
  static Expr Binary(Arena& a, ExprKind k, const Expr& l, const Expr& r) {
    Expr e; e.kind = k;
    e.left = a.make(l);
    e.right = a.make(r);
    return e;
  }
  static Expr Unary(Arena& a, ExprKind k, const Expr& x) {
    Expr e; e.kind = k;
    e.left = a.make(x);
    return e;
  }
  bool is_binary() const { return kind != ExprKind::Var && kind != ExprKind::Constant && kind != ExprKind::BitNot; }
};
static_assert(std::is_trivially_destructible<Expr>::value && std::is_trivially_copyable<Expr>::value, "Expr is plain data");

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

// The `initialized: Vec<bool>` of a Variable (src/ast.rs:67): one flag per element, inline up to 64 elements.
class InitFlags {
 public:
  InitFlags() = default;
  InitFlags(size_t n, bool on) { assign(n, on); }
  size_t size() const { return n_; }
  bool get(size_t i) const {
    if (i >= n_) throw std::out_of_range("InitFlags");
    return (word(i) >> (i & 63)) & 1;
  }
  void set(size_t i, bool on) {
    if (i >= n_) throw std::out_of_range("InitFlags");
    uint64_t& w = i < 64 ? lo_ : more_[i / 64 - 1];
    w = on ? w | (1ull << (i & 63)) : w & ~(1ull << (i & 63));
  }
  void assign(size_t n, bool on) {
    n_ = n;
    lo_ = on ? ~0ull : 0;
    more_.assign(n > 64 ? (n - 1) / 64 : 0, on ? ~0ull : 0);
    trim();
  }
  void push_back(bool on) {
    if (n_ >= 64 && n_ % 64 == 0) more_.push_back(0);
    n_++;
    set(n_ - 1, on);
  }
  size_t count() const {
    size_t c = (size_t)__builtin_popcountll(lo_);
    for (uint64_t w : more_) c += (size_t)__builtin_popcountll(w);
    return c;
  }
  bool all() const { return count() == n_; }
  bool any() const { return count() != 0; }
  struct It {
    const InitFlags* f;
    size_t i;
    bool operator*() const { return f->get(i); }
    It& operator++() { i++; return *this; }
    bool operator!=(const It& o) const { return i != o.i; }
  };
  It begin() const { return {this, 0}; }
  It end() const { return {this, n_}; }

 private:
  uint64_t word(size_t i) const { return i < 64 ? lo_ : more_[i / 64 - 1]; }
  void trim() {  // bits at and above n_ stay 0, so count() needs no mask
    if (n_ < 64) lo_ &= (1ull << n_) - 1;
    if (!more_.empty() && n_ % 64) more_.back() &= (1ull << (n_ % 64)) - 1;
  }
  size_t n_ = 0;
  uint64_t lo_ = 0;
  std::vector<uint64_t> more_;  // elements 64.. (empty for all but long arrays: no allocation)
};

// src/ast.rs:62-88
struct Variable {
  VarRef var_ref;
  std::string name;
  VariableType var_type;
  VariableRole role;
  InitFlags initialized;

  bool all_initialized() const { return initialized.all(); }
  bool any_initialized() const { return initialized.any(); }
  std::vector<size_t> get_initialized() const {
    std::vector<size_t> r;
    for (size_t i = 0; i < initialized.size(); i++) if (initialized.get(i)) r.push_back(i);
    return r;
  }
};

// src/ast.rs:90-111
enum class InstrKind { VarDecl, Assign, ArrayAssign, Constraint, If, For, While };

struct Control;

struct Instruction {
  InstrKind kind = InstrKind::VarDecl;
  VarRef target;                         // Assign / ArrayAssign / Constraint
  Variable var;                          // VarDecl
  const Expr* default_value = nullptr;   // VarDecl (an Option in the Rust: null is None)
  const Expr* index = nullptr;           // ArrayAssign: the index expression
  const Expr* value = nullptr;           // Assign / ArrayAssign / Constraint right-hand side
  const Expr* cond = nullptr;            // If / For / While
  std::shared_ptr<const Control> ctl;    // If / For / While: the blocks (below); null for the other kinds

  // the blocks of a control-flow statement; empty for the other kinds
  inline const std::vector<Instruction>& then_branch() const;                 // If
  inline const std::vector<Instruction>* else_branch() const;                 // If: null when there is none
  inline const Instruction* init() const;                                     // For
  inline const Instruction* step() const;                                     // For
  inline const std::vector<Instruction>& body() const;                        // For / While

  static Instruction VarDecl(Arena& a, Variable v, const std::optional<Expr>& def) {
    Instruction i; i.kind = InstrKind::VarDecl; i.var = std::move(v);
    if (def) i.default_value = a.make(*def);
    return i;
  }
  static Instruction Assign(Arena& a, VarRef t, const Expr& e) {
    Instruction i; i.kind = InstrKind::Assign; i.target = t; i.value = a.make(e); return i;
  }
  static Instruction ArrayAssign(Arena& a, VarRef t, const Expr& idx, const Expr& e) {
    Instruction i; i.kind = InstrKind::ArrayAssign; i.target = t; i.index = a.make(idx); i.value = a.make(e);
    return i;
  }
  static Instruction Constraint(Arena& a, VarRef t, const Expr& e) {
    Instruction i; i.kind = InstrKind::Constraint; i.target = t; i.value = a.make(e); return i;
  }
  static inline Instruction If(Arena& a, const Expr& c, std::vector<Instruction> th,
                               std::optional<std::vector<Instruction>> el);
  static inline Instruction For(Arena& a, Instruction init, const Expr& c, Instruction step,
                                std::vector<Instruction> body);
  static inline Instruction While(Arena& a, const Expr& c, std::vector<Instruction> body);
};

struct Control {
  std::vector<Instruction> then_branch;                 // If
  std::optional<std::vector<Instruction>> else_branch;  // If
  std::optional<Instruction> init, step;                // For
  std::vector<Instruction> body;                        // For / While
};

namespace detail {
inline const std::vector<Instruction>& no_instructions() {
  static const std::vector<Instruction> none;
  return none;
}
}  // namespace detail

const std::vector<Instruction>& Instruction::then_branch() const { return ctl ? ctl->then_branch : detail::no_instructions(); }
const std::vector<Instruction>* Instruction::else_branch() const {
  return ctl && ctl->else_branch ? &*ctl->else_branch : nullptr;
}
const Instruction* Instruction::init() const { return ctl && ctl->init ? &*ctl->init : nullptr; }
const Instruction* Instruction::step() const { return ctl && ctl->step ? &*ctl->step : nullptr; }
const std::vector<Instruction>& Instruction::body() const { return ctl ? ctl->body : detail::no_instructions(); }

Instruction Instruction::If(Arena& a, const Expr& c, std::vector<Instruction> th, std::optional<std::vector<Instruction>> el) {
  Instruction i; i.kind = InstrKind::If; i.cond = a.make(c);
  auto ctl = std::make_shared<Control>();
  ctl->then_branch = std::move(th);
  ctl->else_branch = std::move(el);
  i.ctl = std::move(ctl);
  return i;
}
Instruction Instruction::For(Arena& a, Instruction init, const Expr& c, Instruction step, std::vector<Instruction> body) {
  Instruction i; i.kind = InstrKind::For; i.cond = a.make(c);
  auto ctl = std::make_shared<Control>();
  ctl->init = std::move(init);
  ctl->step = std::move(step);
  ctl->body = std::move(body);
  i.ctl = std::move(ctl);
  return i;
}
Instruction Instruction::While(Arena& a, const Expr& c, std::vector<Instruction> body) {
  Instruction i; i.kind = InstrKind::While; i.cond = a.make(c);
  auto ctl = std::make_shared<Control>();
  ctl->body = std::move(body);
  i.ctl = std::move(ctl);
  return i;
}

// src/ast.rs:113-132
struct Circuit {
  std::vector<Variable> variables;
  std::vector<Instruction> instructions;
  std::shared_ptr<Arena> arena;  // holds every Expr the instructions point to

  Circuit() : arena(std::make_shared<Arena>()) {}
  Circuit(std::vector<Variable> v, std::vector<Instruction> i, std::shared_ptr<Arena> a)
      : variables(std::move(v)), instructions(std::move(i)), arena(std::move(a)) {}

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
