// See fr_host.hpp.  Plain multi-word arithmetic on 64-bit words: products go through a
// word-serial Montgomery reduction (constants are parsed and folded by the thousand when a
// large circuit is lowered); the rare integer operators use bitwise long division.
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
  if (s.size() <= 38) {  // below 10^38 < 2^127 < p: plain 128-bit accumulation, no reduction
    u128 v = 0;
    for (const char ch : s) {
      if (ch < '0' || ch > '9') return false;
      v = v * 10 + (u128)(ch - '0');
    }
    U256 r;
    r.w[0] = (uint64_t)v;
    r.w[1] = (uint64_t)(v >> 64);
    *out = r;
    return true;
  }
  // 18 digits at a time: acc = acc * 10^k + chunk (mod p)
  U256 acc;
  size_t i = 0;
  while (i < s.size()) {
    uint64_t chunk = 0, scale = 1;
    size_t k = 0;
    for (; k < 18 && i < s.size(); k++, i++) {
      const char ch = s[i];
      if (ch < '0' || ch > '9') return false;
      chunk = chunk * 10 + (uint64_t)(ch - '0');
      scale *= 10;
    }
    acc = acc.is_zero() ? U256::from_u64(chunk) : add(mul(acc, U256::from_u64(scale)), U256::from_u64(chunk));
  }
  *out = acc;  // every step kept it below p
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
namespace {
// t * 2^-256 mod p for a 512-bit t < p * 2^256
U256 mont_reduce(U512 t) {
  static const uint64_t np = 0xc2e1f593efffffffull;  // -p^-1 mod 2^64
  const U256& p = modulus();
  uint64_t top = 0;
  for (int i = 0; i < 4; i++) {
    const uint64_t m = t.w[i] * np;
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)m * p.w[j] + t.w[i + j];
      t.w[i + j] = (uint64_t)c;
      c >>= 64;
    }
    for (int k = i + 4; k < 8 && c; k++) {
      c += t.w[k];
      t.w[k] = (uint64_t)c;
      c >>= 64;
    }
    top += (uint64_t)c;
  }
  U256 r, d;
  for (int i = 0; i < 4; i++) r.w[i] = t.w[i + 4];
  if (top || !sub_raw(d, r, p)) {
    sub_raw(d, r, p);
    return d;
  }
  return r;
}
const U256& r2() {  // 2^512 mod p
  static const U256 v = [] {
    U256 x;
    x.w = {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull};
    return x;
  }();
  return v;
}
}  // namespace

U256 mul(const U256& a, const U256& b) {
  // (a*b*R^-1) * R^2 * R^-1 = a*b mod p
  return mont_reduce(mul_wide(mont_reduce(mul_wide(a, b)), r2()));
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
