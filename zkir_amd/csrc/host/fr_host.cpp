// See fr_host.hpp.  Plain multi-word arithmetic: schoolbook products into 512 bits
// and bitwise long division for every reduction; clarity over speed, the lowerer
// folds at most a few thousand constants per circuit.
#include "fr_host.hpp"

#include <algorithm>

namespace zkir {

typedef unsigned __int128 u128;

int cmp(const U256& a, const U256& b) {
  for (int i = 3; i >= 0; i--) {
    if (a.w[i] < b.w[i]) return -1;
    if (a.w[i] > b.w[i]) return 1;
  }
  return 0;
}

namespace {

// r = a + b, returns carry
uint64_t add_raw(U256& r, const U256& a, const U256& b) {
  u128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (u128)a.w[i] + b.w[i];
    r.w[i] = (uint64_t)c;
    c >>= 64;
  }
  return (uint64_t)c;
}
// r = a - b, returns borrow
uint64_t sub_raw(U256& r, const U256& a, const U256& b) {
  uint64_t bw = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a.w[i] - b.w[i] - bw;
    r.w[i] = (uint64_t)d;
    bw = (uint64_t)(d >> 64) & 1;
  }
  return bw;
}

struct U512 {
  uint64_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

U512 mul_wide(const U256& a, const U256& b) {
  U512 r;
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)a.w[i] * b.w[j] + r.w[i + j];
      r.w[i + j] = (uint64_t)c;
      c >>= 64;
    }
    r.w[i + 4] = (uint64_t)c;
  }
  return r;
}

// n mod d and n / d (low 256 bits of the quotient) by restoring division, d != 0
void divmod_wide(const U512& n, const U256& d, U256* q, U256* r) {
  U256 rem, quo;
  for (int bit = 511; bit >= 0; bit--) {
    const uint64_t top = rem.w[3] >> 63;
    for (int i = 3; i > 0; i--) rem.w[i] = (rem.w[i] << 1) | (rem.w[i - 1] >> 63);
    rem.w[0] = (rem.w[0] << 1) | ((n.w[bit >> 6] >> (bit & 63)) & 1);
    U256 t;
    const uint64_t bw = sub_raw(t, rem, d);
    if (top || !bw) {
      rem = t;
      if (bit < 256) quo.w[bit >> 6] |= 1ull << (bit & 63);
    }
  }
  if (q) *q = quo;
  if (r) *r = rem;
}

U512 widen(const U256& a) {
  U512 r;
  for (int i = 0; i < 4; i++) r.w[i] = a.w[i];
  return r;
}

U256 reduce(const U256& a) {  // a mod p for any 256-bit a
  U256 r;
  divmod_wide(widen(a), fr::modulus(), nullptr, &r);
  return r;
}

}  // namespace

namespace fr {

const U256& modulus() {
  static const U256 p = [] {
    U256 v;
    v.w = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    return v;
  }();
  return p;
}

const U256& half() {
  static const U256 h = [] {
    U256 v = modulus();
    for (int i = 0; i < 3; i++) v.w[i] = (v.w[i] >> 1) | (v.w[i + 1] << 63);
    v.w[3] >>= 1;
    return v;
  }();
  return h;
}

bool parse_decimal(const std::string& s, U256* out) {
  if (s.empty()) return false;
  U256 acc;
  const U256 ten = U256::from_u64(10);
  for (char ch : s) {
    if (ch < '0' || ch > '9') return false;
    acc = add(mul(acc, ten), U256::from_u64((uint64_t)(ch - '0')));
  }
  *out = acc;
  return true;
}

std::string to_decimal(const U256& a) {
  if (a.is_zero()) return "0";
  std::string s;
  U256 v = a;
  const U256 ten = U256::from_u64(10);
  while (!v.is_zero()) {
    U256 q, r;
    divmod_wide(widen(v), ten, &q, &r);
    s.push_back((char)('0' + r.w[0]));
    v = q;
  }
  std::reverse(s.begin(), s.end());
  return s;
}

U256 add(const U256& a, const U256& b) {
  U256 s, d;
  add_raw(s, a, b);  // both below p < 2^254: no carry
  return sub_raw(d, s, modulus()) ? s : d;
}
U256 sub(const U256& a, const U256& b) {
  U256 d, e;
  if (!sub_raw(d, a, b)) return d;
  add_raw(e, d, modulus());
  return e;
}
U256 neg(const U256& a) { return a.is_zero() ? a : sub(U256(), a); }
U256 mul(const U256& a, const U256& b) {
  U256 r;
  divmod_wide(mul_wide(a, b), modulus(), nullptr, &r);
  return r;
}
U256 pow(const U256& a, const U256& e) {
  U256 acc = U256::from_u64(1);
  for (int bit = 253; bit >= 0; bit--) {
    acc = mul(acc, acc);
    if (e.bit((unsigned)bit)) acc = mul(acc, a);
  }
  return acc;
}
bool div(const U256& a, const U256& b, U256* out) {
  if (b.is_zero()) return false;
  U256 e;
  sub_raw(e, modulus(), U256::from_u64(2));  // Fermat: b^(p-2)
  *out = mul(a, pow(b, e));
  return true;
}
bool idiv(const U256& a, const U256& b, U256* out) {
  if (b.is_zero()) return false;
  divmod_wide(widen(a), b, out, nullptr);
  return true;
}
bool rem(const U256& a, const U256& b, U256* out) {
  if (b.is_zero()) return false;
  divmod_wide(widen(a), b, nullptr, out);
  return true;
}
bool lt(const U256& a, const U256& b) {
  const bool na = half() < a, nb = half() < b;  // negative representatives
  if (na != nb) return na;
  return a < b;
}
U256 band(const U256& a, const U256& b) {
  U256 r;
  for (int i = 0; i < 4; i++) r.w[i] = a.w[i] & b.w[i];
  return reduce(r);
}
U256 bor(const U256& a, const U256& b) {
  U256 r;
  for (int i = 0; i < 4; i++) r.w[i] = a.w[i] | b.w[i];
  return reduce(r);
}
U256 bxor(const U256& a, const U256& b) {
  U256 r;
  for (int i = 0; i < 4; i++) r.w[i] = a.w[i] ^ b.w[i];
  return reduce(r);
}
U256 bnot254(const U256& a) {
  U256 r;
  for (int i = 0; i < 4; i++) r.w[i] = ~a.w[i];
  r.w[3] &= 0x3fffffffffffffffull;
  return reduce(r);
}
U256 bnot256(const U256& a) {
  U256 r;
  for (int i = 0; i < 4; i++) r.w[i] = ~a.w[i];
  return reduce(r);
}

namespace {
U256 shr_bits(const U256& a, unsigned k) {  // k < 256
  U256 r;
  const unsigned ws = k >> 6, bs = k & 63;
  for (unsigned i = 0; i + ws < 4; i++) {
    uint64_t v = a.w[i + ws] >> bs;
    if (bs && i + ws + 1 < 4) v |= a.w[i + ws + 1] << (64 - bs);
    r.w[i] = v;
  }
  return r;
}
U256 shl_bits_254(const U256& a, unsigned k) {  // (a << k) & (2^254 - 1), k < 256
  U256 r;
  const unsigned ws = k >> 6, bs = k & 63;
  for (unsigned i = ws; i < 4; i++) {
    uint64_t v = a.w[i - ws] << bs;
    if (bs && i - ws >= 1) v |= a.w[i - ws - 1] >> (64 - bs);
    r.w[i] = v;
  }
  r.w[3] &= 0x3fffffffffffffffull;
  return r;
}
// shift amount as a small integer, or -1 when it is 254 or more
int small_amount(const U256& k) {
  if (k.w[1] | k.w[2] | k.w[3]) return -1;
  return k.w[0] >= 254 ? -1 : (int)k.w[0];
}
}  // namespace

U256 shr(const U256& a, const U256& k) {
  if (!(half() < k)) {
    const int s = small_amount(k);
    return s < 0 ? U256() : shr_bits(a, (unsigned)s);
  }
  U256 pk;
  sub_raw(pk, modulus(), k);
  const int s = small_amount(pk);
  return s < 0 ? U256() : reduce(shl_bits_254(a, (unsigned)s));
}
U256 shl(const U256& a, const U256& k) {
  if (!(half() < k)) {
    const int s = small_amount(k);
    return s < 0 ? U256() : reduce(shl_bits_254(a, (unsigned)s));
  }
  U256 pk;
  sub_raw(pk, modulus(), k);
  const int s = small_amount(pk);
  return s < 0 ? U256() : shr_bits(a, (unsigned)s);
}

}  // namespace fr
}  // namespace zkir
