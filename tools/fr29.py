"""Building blocks of BN254-Fr arithmetic on 9 limbs of 29 bits for gfx950, as instruction text.

Why 29 bits: on gfx950 every integer multiply and every add that writes a carry issues at half the rate of
the plain 32-bit operations (v_add_u32, v_sub_u32, v_and_b32, v_or_b32, v_xor_b32, v_lshrrev_b32,
v_ashrrev_i32, v_mov_b32 with register or literal operands).  With 29-bit limbs a column of a product
(at most 18 terms below 2^58 plus a carry) fits the 64-bit accumulator of v_mad_u64_u32, so a Montgomery
product is 162 multiply-adds, 9 low products and 17 64-bit shifts: no carry-writing add at all, against
128 + 8 + 128 of them on 8 limbs of 32 bits.  Additions are 9 plain adds.

Representation: x = sum limb[j] * 2^(29 j); limbs are unsigned 32-bit registers.  "Normal": limbs 0..7
below 2^29.  The Montgomery radix is R = 2^261 (more than 2^7 p), so products of values below 13 p come out
below 2 p without a final subtraction.

Every function appends instruction text to the list `out`.  Register arguments are names ("v12", "s40");
`acc` is the first register number of an even-aligned VGPR pair; `sink` an SGPR pair name that receives the
multiply-adds' carry-out (never read).
"""
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
LIMBS, BITS = 9, 29
MASK = (1 << BITS) - 1
RBITS = LIMBS * BITS  # 261
R = (1 << RBITS) % P
R2 = (R * R) % P
NP = (-pow(P, -1, 1 << BITS)) % (1 << BITS)  # -p^-1 mod 2^29
P_LIMBS = [(P >> (BITS * j)) & MASK for j in range(LIMBS)]


def limbs29(x):
    return [(x >> (BITS * j)) & MASK for j in range(LIMBS - 1)] + [x >> (BITS * (LIMBS - 1))]


def value29(limbs):
    return sum(v << (BITS * j) for j, v in enumerate(limbs))


def pair(n):
    return "v[%d:%d]" % (n, n + 1)


def mul29(out, a, b, r, m, acc, sp, snp, sink, redc_only=False):
    """r = a * b / 2^261 (mod p), limbs of r normal, r < p * (A * B / 2^261 + 1) for a < A, b < B.
    a, b: 9 VGPR names each (limbs may exceed 2^29 as long as 9 * max(a_j) * max(b_j) + 9 * 2^58 + 2^36 < 2^64);
    r: 9 VGPR names, r[j] may be the register of m[j], a[j] or b[j] (each is dead when r[j] is written);
    m: 9 VGPR names for the reduction multipliers; sp: 9 SGPR names holding p's limbs; snp: SGPR with NP.
    redc_only: r = a / 2^261 (mod p): the columns start from a's limbs instead of products (b is ignored)."""
    lo, accp = "v%d" % acc, pair(acc)
    first = True

    def mad(x, y):
        nonlocal first
        out.append("v_mad_u64_u32 %s, %s, %s, %s, %s" % (accp, sink, x, y, "0" if first else accp))
        first = False

    for k in range(2 * LIMBS - 1):
        if redc_only:
            if k < LIMBS:
                if first:  # the accumulator starts as {a_0, 0}
                    out.append("v_mov_b32 %s, %s" % (lo, a[0]))
                    out.append("v_mov_b32 v%d, 0" % (acc + 1))
                    first = False
                else:  # a_k joins the carry: both below 2^36, the sum cannot leave 64 bits
                    mad(a[k], "1")
        else:
            for i in range(max(0, k - LIMBS + 1), min(k, LIMBS - 1) + 1):
                mad(a[i], b[k - i])
        for i in range(max(0, k - LIMBS + 1), min(k - 1, LIMBS - 1) + 1):
            mad(m[i], sp[k - i])
        if k < LIMBS:
            out.append("v_mul_lo_u32 %s, %s, %s" % (m[k], lo, snp))
            out.append("v_and_b32 %s, 0x%x, %s" % (m[k], MASK, m[k]))
            mad(m[k], sp[0])
        else:
            out.append("v_and_b32 %s, 0x%x, %s" % (r[k - LIMBS], MASK, lo))
        out.append("v_lshrrev_b64 %s, %d, %s" % (accp, BITS, accp))
    out.append("v_mov_b32 %s, %s" % (r[LIMBS - 1], lo))


def add29(out, r, a, b):
    """r = a + b limb by limb (no carries move): max(r_j) = max(a_j) + max(b_j) must stay below 2^32."""
    for j in range(LIMBS):
        out.append("v_add_u32 %s, %s, %s" % (r[j], a[j], b[j]))


def sub29(out, r, a, b, c_limbs):
    """r = a - b + C limb by limb; C (limbs c_limbs, literals) is a multiple of p whose every limb is at least
    the bound of b's: nothing goes negative."""
    for j in range(LIMBS):
        out.append("v_add_u32 %s, 0x%x, %s" % (r[j], c_limbs[j], a[j]))
        out.append("v_sub_u32 %s, %s, %s" % (r[j], r[j], b[j]))


def softnorm29(out, x, t):
    """One parallel carry step in place: limbs 0..7 end below 2^29 + 8, limb 8 keeps everything above.  t: one scratch VGPR
    per limb boundary is not needed: the carries are taken from the old limbs top down."""
    # top down so that limb j still holds its old value when limb j+1 takes its carry
    for j in range(LIMBS - 1, 0, -1):
        out.append("v_lshrrev_b32 %s, %d, %s" % (t, BITS, x[j - 1]))
        if j < LIMBS - 1:
            out.append("v_and_b32 %s, 0x%x, %s" % (x[j], MASK, x[j]))
        out.append("v_add_u32 %s, %s, %s" % (x[j], x[j], t))
    out.append("v_and_b32 %s, 0x%x, %s" % (x[0], MASK, x[0]))


def fullnorm29(out, x, t):
    """Sequential carry propagation in place: limbs 0..7 end below 2^29, limb 8 keeps everything above."""
    for j in range(LIMBS - 1):
        out.append("v_lshrrev_b32 %s, %d, %s" % (t, BITS, x[j]))
        out.append("v_and_b32 %s, 0x%x, %s" % (x[j], MASK, x[j]))
        out.append("v_add_u32 %s, %s, %s" % (x[j + 1], x[j + 1], t))


def unpack29(out, r, w, times32=False):
    """8 words of 32 bits (w, a value below 2^256) -> 9 limbs of 29 bits (r); r and w must not overlap.
    times32: the limbs of 32 times the value instead (the value must be below 2^256: the product is below 2^261)."""
    up = 5 if times32 else 0
    for j in range(LIMBS):
        bit = BITS * j - up  # the limb is bits bit .. bit+28 of the value
        if bit < 0:  # the lowest limb of 32 v: v shifted up, the bits below are zero
            out.append("v_lshlrev_b32 %s, %d, %s" % (r[j], -bit, w[0]))
            out.append("v_and_b32 %s, 0x%x, %s" % (r[j], MASK, r[j]))
            continue
        k, s = bit // 32, bit % 32
        if j == LIMBS - 1:
            out.append("v_lshrrev_b32 %s, %d, %s" % (r[j], s, w[k]))  # everything above (bits 232.. or 227..)
        elif s == 0:
            out.append("v_and_b32 %s, 0x%x, %s" % (r[j], MASK, w[k]))
        elif s + BITS <= 32:
            out.append("v_lshrrev_b32 %s, %d, %s" % (r[j], s, w[k]))
            if s + BITS < 32:
                out.append("v_and_b32 %s, 0x%x, %s" % (r[j], MASK, r[j]))
        else:
            out.append("v_alignbit_b32 %s, %s, %s, %d" % (r[j], w[k + 1], w[k], s))
            out.append("v_and_b32 %s, 0x%x, %s" % (r[j], MASK, r[j]))


def pack29(out, w, x, t):
    """9 normal limbs (x, a value below 2^256) -> 8 words of 32 bits (w); w and x must not overlap; t scratch."""
    for k in range(8):
        bit = 32 * k
        j, s = bit // BITS, bit % BITS  # word k starts at bit s of limb j
        # word = (x_j >> s) | (x_{j+1} << (29 - s)) [| (x_{j+2} << (58 - s)) when 58 - s < 32]
        if s == 0:
            out.append("v_lshl_or_b32 %s, %s, %d, %s" % (w[k], x[j + 1], BITS, x[j]))
        else:
            out.append("v_lshrrev_b32 %s, %d, %s" % (t, s, x[j]))
            out.append("v_lshl_or_b32 %s, %s, %d, %s" % (w[k], x[j + 1], BITS - s, t))
        if 2 * BITS - s < 32:
            out.append("v_lshl_or_b32 %s, %s, %d, %s" % (w[k], x[j + 2], 2 * BITS - s, w[k]))
