#pragma once
#include <cstdint>

#include "ast.hpp"

namespace zkir {
// A Poseidon-style permutation chain as a Circuit: `width` input signals, then
// `rounds` rounds of (add round constant, x^5 S-box as three constrained products,
// dense constant mix).  3 * width * rounds constraints, one long dependency chain.
// The constants come from the reference's PRNG (src/rng.rs) seeded with `seed`;
// they are not the parameters of any standard Poseidon instance.
Circuit poseidon_style_gadget(uint32_t width, uint32_t rounds, uint64_t seed);
}  // namespace zkir
