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
    s_ = step(s_);
    return val;
  }

  // The state after n calls of next(), without making them: the update is linear over GF(2), so 2^k steps are one
  // 64 x 64 bit matrix, kept as eight tables of 256 words (one per byte of the state).  The generator draws once
  // per written array before it uses one of the draws (generator.rs:572-601): hundreds of calls whose values
  // nobody reads.
  static uint64_t advanced(uint64_t s, uint64_t n) {
    if (n < 64) {  // (a step is three shifts and three xors; the tables only pay once their cache misses do)
      for (; n; n--) s = step(s);
      return s;
    }
    const JumpTables& t = jump_tables();
    for (int k = 0; n; k++, n >>= 1) {
      if (k >= JumpTables::kPowers) {  // beyond 2^kPowers steps: the rest one table-step at a time
        for (uint64_t r = n; r; r--) s = t.apply(JumpTables::kPowers - 1, t.apply(JumpTables::kPowers - 1, s));
        break;
      }
      if (n & 1) s = t.apply(k, s);
    }
    return s;
  }
  void skip(uint64_t n) { s_ = advanced(s_, n); }
  // what the (k+1)-th call of next() from here would return (the state is left alone)
  uint64_t peek(uint64_t k) const { return advanced(s_, k); }

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
  static uint64_t step(uint64_t s) {
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 43;
    return s;
  }
  struct JumpTables {
    static constexpr int kPowers = 24;
    uint64_t tab[kPowers][8][256];  // tab[k][b][v]: 2^k steps applied to v << (8 b)
    uint64_t apply(int k, uint64_t s) const {
      uint64_t r = 0;
      for (int b = 0; b < 8; b++) r ^= tab[k][b][(s >> (8 * b)) & 0xff];
      return r;
    }
    JumpTables() {
      uint64_t col[64];
      for (int j = 0; j < 64; j++) col[j] = step(1ull << j);
      for (int k = 0; k < kPowers; k++) {
        for (int b = 0; b < 8; b++) {
          tab[k][b][0] = 0;
          for (int v = 1; v < 256; v++) tab[k][b][v] = tab[k][b][v & (v - 1)] ^ col[8 * b + __builtin_ctz((unsigned)v)];
        }
        for (int j = 0; j < 64; j++) col[j] = apply(k, apply(k, 1ull << j));  // the columns of the square
      }
    }
  };
  static const JumpTables& jump_tables() {
    static const JumpTables* t = new JumpTables();  // 393 KB, built once (thread-safe initialisation), never freed
    return *t;
  }
  uint64_t s_;
};

}  // namespace zkir
