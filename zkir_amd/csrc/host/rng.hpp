// The reference's PRNG (src/rng.rs): xorshift64 with shifts 13/17/43 whose next()
// returns the state BEFORE the update, an inclusive modulo-biased rand(min, max)
// that draws nothing when min == max, and helpers built on it.  The only addition
// is a seed constructor: the reference seeds from the timestamp counter
// (rng.rs:14-16) and has no way to replay a stream.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace zkir {

class Rng {
 public:
  // seed 0 is a fixed point of the xorshift and would return 0 forever
  static Rng from_seed(uint64_t seed) {
    if (seed == 0) throw std::invalid_argument("Rng seed must be non-zero");
    return Rng(seed);
  }

  // rng.rs:18-24
  uint64_t next() {
    uint64_t val = s_;
    s_ ^= s_ << 13;
    s_ ^= s_ >> 17;
    s_ ^= s_ << 43;
    return val;
  }

  // rng.rs:26-34
  uint64_t rand(uint64_t min, uint64_t max) {
    if (min > max) throw std::logic_error("Invalid range");
    if (min == max) return min;
    const uint64_t n = max - min + 1;
    if ((n & (n - 1)) == 0 && n) return min + (next() & (n - 1));  // same value as the remainder, without the division
    return min + next() % n;
  }

  // rng.rs:36-45
  char rand_char() {
    static const char kCharset[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz";
    return kCharset[rand(0, sizeof(kCharset) - 2)];
  }
  std::string rand_string(size_t len) {
    std::string s;
    for (size_t i = 0; i < len; i++) s.push_back(rand_char());
    return s;
  }

  // rng.rs:47-52: nullptr stands for None
  template <class T>
  const T* rand_vec(const std::vector<T>& v) {
    if (v.empty()) return nullptr;
    return &v[rand(0, v.size() - 1)];
  }

 private:
  explicit Rng(uint64_t s) : s_(s) {}
  uint64_t s_;
};

}  // namespace zkir
