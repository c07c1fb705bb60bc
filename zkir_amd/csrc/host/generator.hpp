// Random AST producer: the reference's src/generator.rs with its literals lifted
// into a GeneratorConfig whose defaults are those literals, so the default
// configuration consumes the random stream draw for draw like the Rust code.
#pragma once
#include <cstdint>
#include <optional>
#include <vector>

#include "../../../include/zkir_hip.h"
#include "ast.hpp"
#include "ordered_set.hpp"
#include "rng.hpp"

namespace zkir {

using GeneratorConfig = zkir_generator_config;
GeneratorConfig default_generator_config();
// throws std::invalid_argument for a configuration outside what the generator takes
void validate_generator_config(const GeneratorConfig& c);

class Generator {
 public:
  // generator.rs:124-130
  explicit Generator(Rng rng) : rng_(rng), cfg_(default_generator_config()) {}
  Generator(Rng rng, const GeneratorConfig& cfg) : rng_(rng), cfg_(cfg) {}

  // generator.rs:142-164.  A Generator is single use, as in the reference.
  Circuit generate();

  // the two progress lines generate() prints in the reference (generator.rs:148,158)
  size_t requested() const { return requested_; }
  size_t generated() const { return generated_; }

  std::optional<Instruction> generate_instr();                        // generator.rs:166-208
  std::optional<Instruction> gen_var_decl(std::optional<VariableType> type_kind, std::optional<VariableRole> role_kind,
                                          std::optional<Expr> default_value);  // :214-267
  std::optional<Instruction> gen_assign();        // :269-289
  std::optional<Instruction> gen_array_assign();  // :291-327
  std::optional<Instruction> gen_constraint();    // :329-353
  std::optional<Instruction> gen_if();            // :355-399 (not dispatched, as in the reference)
  std::optional<Instruction> gen_for();           // :401-456
  std::optional<Instruction> gen_while();         // :458-483 (not dispatched)
  std::optional<Expr> generate_expr(size_t depth);  // :484-543
  std::optional<Expr> gen_var_ref(size_t depth);    // :545-566
  std::optional<Expr> gen_constant(size_t depth);   // :568-570
  std::optional<Expr> gen_array_index(size_t depth);  // :572-601

 private:
  // generator.rs:30-115.  The reference answers every "which variables can I use here"
  // question by collecting and scanning all variables in scope; that is quadratic in the
  // circuit size.  The same answers, in the same scope order, are kept here incrementally:
  // plain position lists for the static categories, and order-statistic (Fenwick) sets for
  // the ones that change when a variable becomes initialised.
  struct ScopeStack {
    std::vector<Variable> flat;        // every variable in scope, outermost scope first
    std::vector<size_t> scope_begin;   // first position of each open scope
    std::vector<Variable> all_vars;
    size_t next_id = 0;
    // static categories (positions into `flat`, ascending)
    std::vector<size_t> local_fields, local_arrays;
    size_t n_local = 0, n_output = 0;
    // dynamic categories
    OrderedSet ready_fields;     // Field and all_initialized           (gen_var_ref)
    OrderedSet open_outputs;     // Output Field not yet initialised    (gen_constraint)
    OrderedSet touched_arrays;   // Array with any element initialised  (gen_array_index)
    std::vector<size_t> touched_list;     // the members of touched_arrays, ascending
    std::vector<uint32_t> written_count;  // per position: elements initialised so far
    OrderedSet ready_signals;    // signal, Field, all_initialized      (extension: gen_linear)

    void enter_scope() { scope_begin.push_back(flat.size()); }
    void leave_scope();
    bool empty() const { return scope_begin.empty(); }
    Variable add_var(std::string name, VariableType t, VariableRole r, InitFlags initialized);
    void mark_initialized(const VarRef& v);
   private:
    void refresh(size_t pos);
  };

  std::optional<Instruction> gen_var_decl_default() { return gen_var_decl(std::nullopt, std::nullopt, std::nullopt); }
  std::optional<Expr> gen_bin_expr(ExprKind k, size_t depth);
  std::optional<Expr> gen_un_expr(ExprKind k, size_t depth);
  // extensions
  std::optional<Instruction> gen_constraint_quadratic();
  Expr gen_linear(bool factor);
  void push_generated(std::vector<Instruction>& out, std::optional<Instruction>&& instr);

  Rng rng_;
  GeneratorConfig cfg_;
  std::shared_ptr<Arena> arena_ = std::make_shared<Arena>();  // the circuit's expression nodes; handed to the Circuit
  size_t n_instr_ = 0;
  ScopeStack scope_stack_;
  std::vector<Instruction> pending_;  // declarations an extension wants emitted before its statement
  size_t requested_ = 0, generated_ = 0;
};

}  // namespace zkir
