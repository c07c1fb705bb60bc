// Host-side BN254 scalar-field values for the lowerer: parsing the decimal
// `Field` strings of the AST (src/ast.rs:12), folding known sub-expressions and
// deciding loop conditions at lowering time.  It is not an evaluation path: batches
// are only ever evaluated by the GPU kernels.
//
// Operator meanings are circom's, as named by the tokens src/circom.rs:35-157 prints.
#pragma once
#include <array>
#include <cstdint>
#include <string>

namespace zkir {

struct U256 {
  std::array<uint64_t, 4> w{{0, 0, 0, 0}};  // little-endian 64-bit words
  static U256 from_u64(uint64_t v) { U256 r; r.w[0] = v; return r; }
  bool is_zero() const { return (w[0] | w[1] | w[2] | w[3]) == 0; }
  bool bit(unsigned i) const { return (w[i >> 6] >> (i & 63)) & 1; }
  void limbs32(uint32_t out[8]) const {
    for (int i = 0; i < 4; i++) { out[2 * i] = (uint32_t)w[i]; out[2 * i + 1] = (uint32_t)(w[i] >> 32); }
  }
};

int cmp(const U256& a, const U256& b);
inline bool operator==(const U256& a, const U256& b) { return cmp(a, b) == 0; }
inline bool operator<(const U256& a, const U256& b) { return cmp(a, b) < 0; }

namespace fr {

const U256& modulus();
const U256& half();  // floor(p/2)

// decimal text -> value mod p; false if the text is not a non-negative decimal integer
bool parse_decimal(const std::string& s, U256* out);
std::string to_decimal(const U256& a);

U256 add(const U256& a, const U256& b);
U256 sub(const U256& a, const U256& b);
U256 neg(const U256& a);
U256 mul(const U256& a, const U256& b);
U256 pow(const U256& a, const U256& e);
// the three below return false on a zero divisor
bool div(const U256& a, const U256& b, U256* out);
bool idiv(const U256& a, const U256& b, U256* out);
bool rem(const U256& a, const U256& b, U256* out);
bool lt(const U256& a, const U256& b);  // signed representatives
U256 band(const U256& a, const U256& b);
U256 bor(const U256& a, const U256& b);
U256 bxor(const U256& a, const U256& b);
U256 bnot254(const U256& a);
U256 bnot256(const U256& a);
U256 shr(const U256& a, const U256& k);
U256 shl(const U256& a, const U256& k);

}  // namespace fr
}  // namespace zkir
