// Host code only (no GPU): how fast one thread generates and lowers campaign circuits (the configuration
// zkir_amd/campaign.py::campaign_config draws: straight-line arithmetic circuits, 64 to 4 096 constraints,
// log-uniform by circuit id).  Prints milliseconds per circuit for generate, lower and both through
// zkir_generate_lower_many.  Build line in README.txt; with -pg it is the input of gprof.
#include "../../include/zkir_hip.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 512;
  const uint32_t threads = argc > 2 ? (uint32_t)atoi(argv[2]) : 1;
  std::vector<uint64_t> seeds(n);
  std::vector<zkir_generator_config> cfgs(n);
  const double lo = std::log(64 * 4.0), hi = std::log(4096 * 4.0);
  for (uint32_t k = 0; k < n; k++) {
    seeds[k] = 7 + k + 1;
    zkir_generator_config_default(&cfgs[k]);
    const double u = (double)(((uint64_t)k * 2654435761ull) % 4096) / 4095.0;
    const uint32_t m = (uint32_t)std::lround(std::exp(lo + (hi - lo) * u));
    cfgs[k].min_instructions = cfgs[k].max_instructions = m;
    cfgs[k].forced_inputs = 8;
    cfgs[k].inputs_readable = cfgs[k].arithmetic_only = cfgs[k].quadratic_aware = 1;
    cfgs[k].no_loops = cfgs[k].safe_array_index = 1;
  }
  zkir_lower_options o;
  zkir_lower_options_default(&o);
  o.reorder = o.order_ops = o.window_cache = o.scaled_values = 0;  // as a campaign lowers (zkir_amd/campaign.py::campaign_lower_options)
  const bool only_many = argc > 3;  // a third argument: time only zkir_generate_lower_many
  double t0 = now();
  std::vector<zkir_circuit*> cs(only_many ? 0 : n);
  for (uint32_t k = 0; k < cs.size(); k++)
    if (zkir_generate(seeds[k], &cfgs[k], &cs[k])) return 1;
  double t1 = now();
  size_t constraints = 0, refused = 0;
  for (uint32_t k = 0; k < cs.size(); k++) {
    zkir_tape* t = nullptr;
    int32_t cls = 0;
    const int rc = zkir_lower(cs[k], &o, &t, &cls);
    if (rc == ZKIR_E_LOWER) { refused++; continue; }
    if (rc) return 2;
    zkir_tape_hdr h;
    zkir_tape_header(t, &h);
    constraints += h.n_constraints;
    zkir_tape_free(t);
  }
  double t2 = now();
  for (zkir_circuit* c : cs) zkir_circuit_free(c);
  std::vector<zkir_tape*> tapes(n);
  std::vector<int32_t> classes(n);
  double t3 = now();
  if (zkir_generate_lower_many(seeds.data(), cfgs.data(), n, &o, threads, tapes.data(), classes.data())) return 3;
  double t4 = now();
  for (zkir_tape* t : tapes) zkir_tape_free(t);
  printf("%u circuits, %zu refused, %.0f constraints each: generate %.3f ms, lower %.3f ms per circuit (one thread); "
         "generate_lower_many on %u threads %.3f ms per circuit = %.0f circuits/s\n",
         n, refused, only_many ? 0.0 : (double)constraints / (n - refused), 1e3 * (t1 - t0) / n, 1e3 * (t2 - t1) / n, threads,
         1e3 * (t4 - t3) / n, n / (t4 - t3));
  return 0;
}
