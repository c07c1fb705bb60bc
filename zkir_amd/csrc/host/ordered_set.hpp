// A set of positions 0..n-1 that answers "how many members" and "which is the k-th member" in O(log n): a
// Fenwick tree over membership flags.  The generator asks such questions for every random choice among "the
// variables in scope that ..." (the reference collects and scans all variables each time, src/generator.rs:30-115).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace zkir {

class OrderedSet {
 public:
  void resize(size_t n) {  // positions 0..n-1 exist; new ones are absent
    if (n < flag_.size()) {
      for (size_t p = n; p < flag_.size(); p++) set(p, false);
      flag_.resize(n);
      return;
    }
    flag_.resize(n, 0);
    if (n + 1 > tree_.size()) {  // grow to the next power of two and rebuild
      size_t cap = 16;
      while (cap < n + 1) cap *= 2;
      tree_.assign(cap, 0);
      for (size_t p = 0; p < flag_.size(); p++)
        if (flag_[p])
          for (size_t i = p + 1; i < tree_.size(); i += i & (~i + 1)) tree_[i] += 1;
    }
  }
  void set(size_t pos, bool on) {
    if (pos >= flag_.size() || (bool)flag_[pos] == on) return;
    flag_[pos] = on;
    count_ += on ? 1 : (size_t)-1;
    for (size_t i = pos + 1; i < tree_.size(); i += i & (~i + 1)) tree_[i] += on ? 1 : -1;
  }
  bool test(size_t pos) const { return pos < flag_.size() && flag_[pos]; }
  size_t count() const { return count_; }
  size_t select(size_t k) const {  // position of the k-th (0-based) member
    size_t pos = 0;
    int64_t rem = (int64_t)k + 1;
    for (size_t step = tree_.size() / 2; step > 0; step /= 2)
      if (pos + step < tree_.size() && tree_[pos + step] < rem) {
        pos += step;
        rem -= tree_[pos];
      }
    return pos;
  }

 private:
  std::vector<uint8_t> flag_;
  std::vector<int32_t> tree_;  // Fenwick tree over flag_, 1-based, capacity a power of two
  size_t count_ = 0;
};

}  // namespace zkir
