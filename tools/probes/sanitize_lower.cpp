// Host code only (no GPU): generate, lower under every combination of the placement options (plain tapes in program
// order and reordered, window as a cache or not, strand tapes of four and eight strands), validate, print and parse;
// scaled values on and off, circuits with and without constant factors; and the batched generate + lower on several
// threads, both as generate-then-lower and fused (campaign configurations).  Built with -fsanitize=address,undefined (README.txt).
#include "../../include/zkir_hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

static size_t strand_tapes = 0, total = 0, numbered = 0;

static int lower_and_validate(zkir_circuit* c, const zkir_lower_options& o) {
  zkir_tape* t = nullptr;
  int32_t cls = 0;
  const int rc = zkir_lower(c, &o, &t, &cls);
  if (rc == ZKIR_E_LOWER) return 0;  // refused circuits are a result
  if (rc) { printf("lower failed: %s\n", zkir_host_last_error()); return 1; }
  zkir_tape_hdr h;
  zkir_tape_header(t, &h);
  strand_tapes += h.flags & 1;
  total++;
  const void* blob;
  size_t bytes;
  zkir_tape_blob(t, &blob, &bytes);
  const zkir_op* ops = (const zkir_op*)blob;
  for (uint32_t i = 0; i < h.n_ops && !(h.flags & 1); i++)
    if ((ops[i].code & 0xff) == ZKIR_OP_CHECK && (ops[i].code >> 8)) { numbered++; break; }
  zkir_tape* t2 = nullptr;
  if (zkir_tape_from_blob(&h, blob, bytes, &t2)) { printf("validation failed: %s\n", zkir_host_last_error()); return 1; }
  zkir_tape_free(t2);
  zkir_tape_free(t);
  return 0;
}

static int all_options(zkir_circuit* c) {
  for (uint32_t strands : {0u, 4u, 8u})
    for (uint32_t reorder : {0u, 1u})
      for (uint32_t cache : {0u, 1u})
        for (uint32_t rows : {0u, 2u, 16u}) {
          zkir_lower_options o;
          zkir_lower_options_default(&o);
          o.strands = strands; o.reorder = reorder; o.window_cache = cache; o.order_ops = (rows >> 1) & 1u; o.n_lslots = rows; o.max_iterations = 300;
          o.scaled_values = (rows >> 4) ^ reorder;  // both settings meet every other option
          if (lower_and_validate(c, o)) return 1;
        }
  return 0;
}

int main() {
  for (int w = 1; w <= 6; w++)
    for (int r = 1; r <= 40; r += 13) {
      zkir_circuit* c = nullptr;
      if (zkir_gadget_poseidon_style(w, r, 7 + w, &c)) return 1;
      if (all_options(c)) return 1;
      zkir_circuit_free(c);
    }
  for (int seed = 1; seed <= 160; seed++) {
    zkir_generator_config g;
    zkir_generator_config_default(&g);
    g.min_instructions = g.max_instructions = 10 + seed * 5;
    g.forced_inputs = 4; g.inputs_readable = 1; g.arithmetic_only = seed % 3 != 0; g.quadratic_aware = 1;
    g.no_loops = seed % 2; g.safe_array_index = 1; g.max_linear_terms = 1 + seed % 2; g.enable_if = seed % 5 == 0 ? 2 : 0;
    g.signal_factors = seed % 4 == 1;
    zkir_circuit* c = nullptr;
    if (zkir_generate(seed, &g, &c)) return 1;
    if (all_options(c)) return 1;
    // the text form, there and back
    size_t need = 0;
    zkir_circuit_print(c, nullptr, 0, &need);
    std::vector<char> text(need);
    if (zkir_circuit_print(c, text.data(), text.size(), &need)) return 1;
    zkir_circuit* c2 = nullptr;
    if (zkir_circuit_parse(text.data(), &c2)) { printf("parse failed: %s\n", zkir_host_last_error()); return 1; }
    zkir_circuit_transpile_circom(c2, nullptr, 0, &need);
    zkir_circuit_free(c2);
    zkir_circuit_free(c);
  }
  // batched generate + lower on four threads, one bad seed among good ones in a second call
  std::vector<uint64_t> seeds;
  for (uint64_t s = 1; s <= 64; s++) seeds.push_back(s * 977);
  std::vector<zkir_tape*> tapes(seeds.size());
  std::vector<int32_t> classes(seeds.size());
  if (zkir_generate_lower_many(seeds.data(), nullptr, (uint32_t)seeds.size(), nullptr, 4, tapes.data(), classes.data())) return 1;
  for (zkir_tape* t : tapes) zkir_tape_free(t);
  seeds[10] = 0;
  if (zkir_generate_lower_many(seeds.data(), nullptr, (uint32_t)seeds.size(), nullptr, 4, tapes.data(), classes.data()) != ZKIR_E_INVALID) return 1;
  for (zkir_tape* t : tapes) if (t) { printf("tape left behind by a failed call\n"); return 1; }
  // campaign configurations (what the fused generate + lower takes), program-order lowering and the full placement
  seeds[10] = 10 * 977;
  std::vector<zkir_generator_config> cfgs(seeds.size());
  for (size_t k = 0; k < cfgs.size(); k++) {
    zkir_generator_config& g = cfgs[k];
    zkir_generator_config_default(&g);
    g.min_instructions = g.max_instructions = 40 + 37 * (uint32_t)k;
    g.forced_inputs = 8; g.inputs_readable = 1; g.arithmetic_only = 1; g.quadratic_aware = 1; g.no_loops = 1; g.safe_array_index = 1;
    g.signal_factors = k % 3 == 0; g.max_linear_terms = 1 + k % 2;
  }
  for (int full = 0; full < 2; full++) {
    zkir_lower_options o;
    zkir_lower_options_default(&o);
    if (!full) o.reorder = o.order_ops = o.window_cache = 0;
    if (zkir_generate_lower_many(seeds.data(), cfgs.data(), (uint32_t)seeds.size(), &o, 4, tapes.data(), classes.data())) return 1;
    for (zkir_tape* t : tapes) {
      if (!t) continue;
      zkir_tape_hdr h;
      zkir_tape_header(t, &h);
      const void* blob;
      size_t bytes;
      zkir_tape_blob(t, &blob, &bytes);
      zkir_tape* t2 = nullptr;
      if (zkir_tape_from_blob(&h, blob, bytes, &t2)) { printf("validation failed: %s\n", zkir_host_last_error()); return 1; }
      zkir_tape_free(t2);
      zkir_tape_free(t);
      total++;
    }
  }
  printf("lowered %zu tapes (%zu strand tapes, %zu reordered plain tapes), all validate; text form and batched call clean\n",
         total, strand_tapes, numbered);
}
