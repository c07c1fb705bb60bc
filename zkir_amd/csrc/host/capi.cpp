// The host half of include/zkir_hip.h: the reference-side mirror (generate, print,
// transpile) and lower, behind a C boundary that never lets an exception through.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/zkir_hip.h"
#include "../hip/tape_check.h"
#include "ast.hpp"
#include "circom.hpp"
#include "noir.hpp"
#include "gadget.hpp"
#include "generator.hpp"
#include "lower.hpp"
#include "rng.hpp"
#include "sexpr.hpp"

namespace {

thread_local std::string g_host_error;

// thread-nanoseconds zkir_generate_lower_many has spent generating and lowering, over all its calls and threads
std::atomic<uint64_t> g_generate_ns{0}, g_lower_ns{0};
inline uint64_t now_ns() {
  return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int32_t copy_out(const std::string& s, char* buf, size_t cap, size_t* needed) {
  if (needed) *needed = s.size() + 1;
  if (!buf || cap < s.size() + 1) return ZKIR_E_BUFSIZE;
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return ZKIR_OK;
}

template <class F>
int32_t guarded(F f) {
  try {
    return f();
  } catch (const zkir::LowerError& e) {
    g_host_error = e.what();
    return ZKIR_E_LOWER;
  } catch (const zkir::NoirUnsupported& e) {
    g_host_error = e.what();
    return ZKIR_E_UNSUPPORTED;
  } catch (const std::bad_alloc&) {
    g_host_error = "out of memory";
    return ZKIR_E_NOMEM;
  } catch (const std::invalid_argument& e) {
    g_host_error = e.what();
    return ZKIR_E_INVALID;
  } catch (const std::exception& e) {
    g_host_error = e.what();
    return ZKIR_E_INTERNAL;
  } catch (...) {
    g_host_error = "unknown exception";
    return ZKIR_E_INTERNAL;
  }
}

}  // namespace

extern "C" {

// message of the last failing host-side call on this thread
const char* zkir_host_last_error(void) { return g_host_error.c_str(); }

int32_t zkir_generator_config_default(zkir_generator_config* cfg) {
  if (!cfg) return ZKIR_E_INVALID;
  *cfg = zkir::default_generator_config();
  return ZKIR_OK;
}

int32_t zkir_generate(uint64_t seed, const zkir_generator_config* cfg, zkir_circuit** out) {
  if (!out) return ZKIR_E_INVALID;
  *out = nullptr;
  return guarded([&]() -> int32_t {
    zkir::GeneratorConfig c = cfg ? *cfg : zkir::default_generator_config();
    zkir::validate_generator_config(c);
    zkir::Generator g(zkir::Rng::from_seed(seed), c);
    auto* h = new zkir_circuit();
    h->circuit = g.generate();
    h->requested = g.requested();
    h->generated = g.generated();
    *out = h;
    return ZKIR_OK;
  });
}

int32_t zkir_circuit_generation_counts(const zkir_circuit* c, uint64_t* requested, uint64_t* generated) {
  if (!c || !requested || !generated) return ZKIR_E_INVALID;
  *requested = c->requested;
  *generated = c->generated;
  return ZKIR_OK;
}

int32_t zkir_rng_stream(uint64_t seed, uint32_t n, uint64_t* out_vals) {
  if (!out_vals && n) return ZKIR_E_INVALID;
  return guarded([&]() -> int32_t {
    zkir::Rng r = zkir::Rng::from_seed(seed);
    for (uint32_t i = 0; i < n; i++) out_vals[i] = r.next();
    return ZKIR_OK;
  });
}

int32_t zkir_circuit_parse(const char* text, zkir_circuit** out) {
  if (!text || !out) return ZKIR_E_INVALID;
  *out = nullptr;
  return guarded([&]() -> int32_t {
    zkir::Circuit c;
    try {
      c = zkir::parse_circuit(text);
    } catch (const std::runtime_error& e) {
      throw std::invalid_argument(e.what());
    }
    auto* h = new zkir_circuit();
    h->circuit = std::move(c);
    *out = h;
    return ZKIR_OK;
  });
}

int32_t zkir_circuit_print(const zkir_circuit* c, char* buf, size_t cap, size_t* needed) {
  if (!c) return ZKIR_E_INVALID;
  return guarded([&]() -> int32_t { return copy_out(zkir::print_circuit(c->circuit), buf, cap, needed); });
}

int32_t zkir_circuit_transpile_circom(const zkir_circuit* c, char* buf, size_t cap, size_t* needed) {
  if (!c) return ZKIR_E_INVALID;
  return guarded([&]() -> int32_t {
    zkir::CircomTranspiler t(c->circuit);
    return copy_out(t.transpile_circuit(), buf, cap, needed);
  });
}

int32_t zkir_circuit_transpile_noir(const zkir_circuit* c, char* buf, size_t cap, size_t* needed) {
  if (!c) return ZKIR_E_INVALID;
  return guarded([&]() -> int32_t {
    zkir::NoirTranspiler t(c->circuit);
    return copy_out(t.transpile_circuit(), buf, cap, needed);
  });
}

int32_t zkir_circuit_debug(const zkir_circuit* c, char* buf, size_t cap, size_t* needed) {
  if (!c) return ZKIR_E_INVALID;
  return guarded([&]() -> int32_t { return copy_out(zkir::debug_instructions(c->circuit.instructions), buf, cap, needed); });
}

void zkir_circuit_free(zkir_circuit* c) { delete c; }

int32_t zkir_gadget_poseidon_style(uint32_t width, uint32_t rounds, uint64_t seed, zkir_circuit** out) {
  if (!out) return ZKIR_E_INVALID;
  *out = nullptr;
  return guarded([&]() -> int32_t {
    if ((uint64_t)width * rounds > (1u << 22)) throw std::invalid_argument("gadget too large");
    auto* h = new zkir_circuit();
    h->circuit = zkir::poseidon_style_gadget(width, rounds, seed);
    *out = h;
    return ZKIR_OK;
  });
}

int32_t zkir_lower_options_default(zkir_lower_options* o) {
  if (!o) return ZKIR_E_INVALID;
  *o = zkir::default_lower_options();
  return ZKIR_OK;
}

int32_t zkir_lower(const zkir_circuit* c, const zkir_lower_options* opts, zkir_tape** out, int32_t* status_class) {
  if (!c || !out) return ZKIR_E_INVALID;
  *out = nullptr;
  if (status_class) *status_class = ZKIR_LOWER_OK;
  try {
    zkir::LowerOptions o = opts ? *opts : zkir::default_lower_options();
    zkir::Lowerer lw(c->circuit, o);
    auto* h = new zkir_tape();
    try {
      h->tape = lw.lower();
    } catch (...) {
      delete h;
      throw;
    }
    *out = h;
    return ZKIR_OK;
  } catch (const zkir::LowerError& e) {
    g_host_error = e.what();
    if (status_class) *status_class = e.status_class;
    return ZKIR_E_LOWER;
  } catch (const std::bad_alloc&) {
    g_host_error = "out of memory";
    return ZKIR_E_NOMEM;
  } catch (const std::exception& e) {
    g_host_error = e.what();
    return ZKIR_E_INTERNAL;
  } catch (...) {
    g_host_error = "unknown exception";
    return ZKIR_E_INTERNAL;
  }
}

// A campaign's inner loop on the host: circuit k is what zkir_generate(seeds[k], &cfgs[k]) followed by
// zkir_lower(., opts) would give, done on `threads` host threads with nothing crossing the boundary per
// circuit.  tapes[k] is the tape (status_classes[k] == ZKIR_LOWER_OK) or NULL with the class that refused it.
int32_t zkir_generate_lower_many(const uint64_t* seeds, const zkir_generator_config* cfgs, uint32_t n,
                                 const zkir_lower_options* opts, uint32_t threads, zkir_tape** tapes,
                                 int32_t* status_classes) {
  if (n && (!seeds || !tapes || !status_classes)) return ZKIR_E_INVALID;
  for (uint32_t k = 0; k < n; k++) {
    tapes[k] = nullptr;
    status_classes[k] = ZKIR_LOWER_OK;
  }
  const zkir::LowerOptions o = opts ? *opts : zkir::default_lower_options();
  const bool fused = std::getenv("ZKIR_NO_FUSED") == nullptr;  // test aid: set, every circuit is built and then lowered
  std::atomic<uint32_t> next{0};
  std::atomic<int32_t> failed{ZKIR_OK};
  std::mutex error_lock;
  std::string first_error;
  auto work = [&]() {
    uint64_t gen_ns = 0, low_ns = 0;
    struct Tally {  // added to the process-wide counters when the worker ends, however it ends
      uint64_t &g, &l;
      ~Tally() {
        g_generate_ns.fetch_add(g);
        g_lower_ns.fetch_add(l);
      }
    } tally{gen_ns, low_ns};
    for (;;) {
      const uint32_t k = next.fetch_add(1);
      if (k >= n || failed.load() != ZKIR_OK) return;
      const zkir::GeneratorConfig cfg = cfgs ? cfgs[k] : zkir::default_generator_config();
      int32_t rc = ZKIR_OK;
      const uint64_t t0 = now_ns();
      if (fused && zkir::can_lower_generated(cfg, o)) {
        // the circuit is never built: its statements go to pass 1 as they are drawn (same draws, same tape)
        double pass1_s = 0;
        rc = guarded([&]() -> int32_t {
          auto* h = new zkir_tape();
          try {
            h->tape = zkir::lower_generated(seeds[k], cfg, o, &pass1_s);
          } catch (const zkir::LowerError& e) {
            delete h;
            status_classes[k] = e.status_class;
            return ZKIR_OK;  // a refused circuit is a result, not a failure of the call
          } catch (...) {
            delete h;
            throw;
          }
          tapes[k] = h;
          return ZKIR_OK;
        });
        const uint64_t t1 = now_ns();
        const uint64_t p1 = std::min<uint64_t>((uint64_t)(pass1_s * 1e9), t1 - t0);
        gen_ns += p1;            // drawing the circuit and pass 1
        low_ns += t1 - t0 - p1;  // slot placement
      } else {
        zkir_circuit* c = nullptr;
        rc = zkir_generate(seeds[k], &cfg, &c);
        const uint64_t t1 = now_ns();
        gen_ns += t1 - t0;
        if (rc == ZKIR_OK) {
          rc = zkir_lower(c, &o, &tapes[k], &status_classes[k]);
          if (rc == ZKIR_E_LOWER) rc = ZKIR_OK;  // a refused circuit is a result, not a failure of the call
        }
        zkir_circuit_free(c);
        low_ns += now_ns() - t1;
      }
      if (rc != ZKIR_OK) {
        std::lock_guard<std::mutex> g(error_lock);
        if (failed.load() == ZKIR_OK) {
          failed.store(rc);
          first_error = "circuit " + std::to_string(k) + ": " + g_host_error;  // this worker's own message
        }
        return;
      }
    }
  };
  // The pool lives outside the try block: when a thread cannot be started (std::system_error at the thread limit)
  // the ones that did start are joined, not destroyed while joinable (that would end the process), and the
  // circuits are made by the threads there are.
  std::vector<std::thread> pool;
  try {
    const uint32_t n_threads = std::max<uint32_t>(1, std::min<uint32_t>(std::min<uint32_t>(threads ? threads : 1, n), 256));
    pool.reserve(n_threads);
    try {
      for (uint32_t t = 1; t < n_threads; t++) pool.emplace_back(work);
    } catch (const std::system_error&) {
    }
    work();
  } catch (const std::exception& e) {
    std::lock_guard<std::mutex> g(error_lock);
    failed.store(ZKIR_E_INTERNAL);
    first_error = e.what();
  } catch (...) {
    std::lock_guard<std::mutex> g(error_lock);
    failed.store(ZKIR_E_INTERNAL);
    first_error = "unknown exception";
  }
  for (std::thread& t : pool)
    if (t.joinable()) t.join();
  if (failed.load() != ZKIR_OK) {
    for (uint32_t k = 0; k < n; k++) {
      zkir_tape_free(tapes[k]);
      tapes[k] = nullptr;
    }
    g_host_error = first_error;
  }
  return failed.load();
}

int32_t zkir_host_phase_seconds(double* generate_s, double* lower_s) {
  if (!generate_s || !lower_s) return ZKIR_E_INVALID;
  *generate_s = 1e-9 * (double)g_generate_ns.load();
  *lower_s = 1e-9 * (double)g_lower_ns.load();
  return ZKIR_OK;
}

int32_t zkir_tape_header(const zkir_tape* t, zkir_tape_hdr* hdr) {
  if (!t || !hdr) return ZKIR_E_INVALID;
  *hdr = t->tape.hdr;
  return ZKIR_OK;
}

int32_t zkir_tape_blob(const zkir_tape* t, const void** blob, size_t* bytes) {
  if (!t || !blob || !bytes) return ZKIR_E_INVALID;
  *blob = t->tape.blob.data();
  *bytes = t->tape.blob.size();
  return ZKIR_OK;
}

// (var id, offset, len) triples of the input (which = 0) or output (which = 1) signals
int32_t zkir_tape_io_map(const zkir_tape* t, int32_t which, uint64_t* triples, size_t cap_entries, size_t* n_entries) {
  if (!t || !n_entries) return ZKIR_E_INVALID;
  const auto& v = which ? t->tape.outputs : t->tape.inputs;
  *n_entries = v.size();
  if (!triples || cap_entries < v.size()) return ZKIR_E_BUFSIZE;
  for (size_t i = 0; i < v.size(); i++) {
    triples[3 * i] = v[i].var_id;
    triples[3 * i + 1] = v[i].offset;
    triples[3 * i + 2] = v[i].len;
  }
  return ZKIR_OK;
}

int32_t zkir_tape_from_blob(const zkir_tape_hdr* hdr, const void* blob, size_t bytes, zkir_tape** out) {
  if (!hdr || !out) return ZKIR_E_INVALID;
  *out = nullptr;
  return guarded([&]() -> int32_t {
    std::string why;
    if (!zkir::check_tape(*hdr, blob, bytes, &why)) throw std::invalid_argument("bad tape: " + why);
    auto* h = new zkir_tape();
    h->tape.hdr = *hdr;
    h->tape.blob.assign((const uint8_t*)blob, (const uint8_t*)blob + bytes);
    *out = h;
    return ZKIR_OK;
  });
}

void zkir_tape_free(zkir_tape* t) { delete t; }

}  // extern "C"
