#!/usr/bin/env python3
"""Emit zkir_amd/csrc/hip/fr_mul_asm.inc: the gfx950 instruction stream of one
BN254-Fr Montgomery multiplication (8 x 32-bit limbs, R = 2^256).

The stream is CIOS with the 64-bit multiply-add (v_mad_u64_u32: a*b + c64):
  row i:  D_j = a_j*b_i + {t_j,0}      8 independent multiply-adds
          t   = lo(D) + (hi(D) << 32)  one carry chain
          m   = t_0 * (-p^-1 mod 2^32)
          D_j = m*p_j + {t_j,0}        8 independent multiply-adds
          t   = (lo(D) + (hi(D) << 32)) >> 32   one carry chain
so each 32x32 product costs one multiply-add and one add-with-carry.  A carry
written by one VALU instruction may be consumed only after one other
instruction has issued, so the two chains of a row are interleaved with the
multiply-adds that follow them.

Register plan (physical VGPRs TB..TB+33, clobbered by the asm statement):
  E_j = v[TB+2j]   running value t_j        Z_j = v[TB+2j+1] = 0   (addend pairs)
  L_j = v[TB+16+2j], H_j = v[TB+17+2j]      product pairs
  E_8 = v[TB+32], M = v[TB+33]
"""
import sys

TB = int(sys.argv[1]) if len(sys.argv) > 1 else 40
OUT = sys.argv[2] if len(sys.argv) > 2 else "zkir_amd/csrc/hip/fr_mul_asm.inc"

# operand numbering of the asm statement:
#   %0..%7   r (outputs, early clobber)
#   %8       SGPR pair that receives the multiply-adds' carry-out (an OUTPUT: the
#            instruction writes it, so the compiler must not keep anything there)
#   %9..%16  a
#   %17..%24 b
#   %25..%32 p (VGPR: the borrow chain of the final subtraction already uses the
#            one scalar operand a VALU instruction may read)
#   %33      np0 (SGPR)
R = lambda j: "%%%d" % j
A = lambda j: "%%%d" % (9 + j)
B = lambda j: "%%%d" % (17 + j)
P = lambda j: "%%%d" % (25 + j)
NP = "%33"
E = lambda j: "v%d" % (TB + 2 * j) if j < 8 else "v%d" % (TB + 32)
Z = lambda j: "v%d" % (TB + 2 * j + 1)
EZ = lambda j: "v[%d:%d]" % (TB + 2 * j, TB + 2 * j + 1)
L = lambda j: "v%d" % (TB + 16 + 2 * j)
H = lambda j: "v%d" % (TB + 17 + 2 * j)
D = lambda j: "v[%d:%d]" % (TB + 16 + 2 * j, TB + 17 + 2 * j)
M = "v%d" % (TB + 33)

lines = []
emit = lines.append


def mad(dst, x, y, addend):
    emit("v_mad_u64_u32 %s, %%8, %s, %s, %s" % (dst, x, y, addend))


for j in range(8):
    emit("v_mov_b32 %s, 0" % Z(j))
for i in range(8):
    if i == 0:
        for j in range(8):
            mad(D(j), A(j), B(0), "0")
    # chain 1 (t += a*b_i), interleaved with the m*p multiply-adds
    emit("v_mov_b32 %s, %s" % (E(0), L(0)))
    emit("v_add_co_u32 %s, vcc, %s, %s" % (E(1), L(1), H(0)))
    emit("v_mul_lo_u32 %s, %s, %s" % (M, E(0), NP))
    for j in range(2, 9):
        if j < 8:
            emit("v_addc_co_u32 %s, vcc, %s, %s, vcc" % (E(j), L(j), H(j - 1)))
        else:
            emit("v_addc_co_u32 %s, vcc, 0, %s, vcc" % (E(8), H(7)))
        mad(D(j - 2), M, P(j - 2), EZ(j - 2))
    mad(D(7), M, P(7), EZ(7))
    # chain 2 ((t + m*p) >> 32), interleaved with the next row's a*b multiply-adds
    emit("v_add_co_u32 %s, vcc, %s, %s" % (E(0), L(1), H(0)))
    if i < 7:
        mad(D(0), A(0), B(i + 1), EZ(0))
    else:
        emit("s_nop 0")
    for j in range(2, 9):
        if j < 8:
            emit("v_addc_co_u32 %s, vcc, %s, %s, vcc" % (E(j - 1), L(j), H(j - 1)))
        else:
            emit("v_addc_co_u32 %s, vcc, %s, %s, vcc" % (E(7), E(8), H(7)))
        if i < 7:
            mad(D(j - 1), A(j - 1), B(i + 1), EZ(j - 1))
        elif j < 8:
            emit("s_nop 0")
# final conditional subtraction: d = t - p; borrow set <=> t < p
emit("v_sub_co_u32 %s, vcc, %s, %s" % (L(0), E(0), P(0)))
for j in range(1, 8):
    emit("s_nop 0")
    emit("v_subb_co_u32 %s, vcc, %s, %s, vcc" % (L(j), E(j), P(j)))
emit("s_nop 1")
for j in range(8):
    emit("v_cndmask_b32 %s, %s, %s, vcc" % (R(j), L(j), E(j)))


# ---- Montgomery reduction alone: r = a * 2^-256 mod p for a < p (leaving Montgomery form).
# Same registers; operands: %0..%7 r, %8 carry sink, %9..%16 a, %17..%24 p, %25 np0.
RA = lambda j: "%%%d" % (9 + j)
RP = lambda j: "%%%d" % (17 + j)
RNP = "%25"
rl = []
for j in range(8):
    rl.append("v_mov_b32 %s, 0" % Z(j))
for j in range(8):
    rl.append("v_mov_b32 %s, %s" % (E(j), RA(j)))
rl.append("v_mul_lo_u32 %s, %s, %s" % (M, E(0), RNP))
for j in range(8):
    rl.append("v_mad_u64_u32 %s, %%8, %s, %s, %s" % (D(j), M, RP(j), EZ(j)))
for i in range(8):
    last = i == 7
    rl.append("v_add_co_u32 %s, vcc, %s, %s" % (E(0), L(1), H(0)))
    rl.append("s_nop 0" if last else "v_mul_lo_u32 %s, %s, %s" % (M, E(0), RNP))
    for j in range(2, 9):
        if j < 8:
            rl.append("v_addc_co_u32 %s, vcc, %s, %s, vcc" % (E(j - 1), L(j), H(j - 1)))
        else:
            rl.append("v_addc_co_u32 %s, vcc, 0, %s, vcc" % (E(7), H(7)))
        if not last:
            rl.append("v_mad_u64_u32 %s, %%8, %s, %s, %s" % (D(j - 2), M, RP(j - 2), EZ(j - 2)))
        elif j < 8:
            rl.append("s_nop 0")
    if not last:
        rl.append("v_mad_u64_u32 %s, %%8, %s, %s, %s" % (D(7), M, RP(7), EZ(7)))
# a < p makes the result < p: no final subtraction
for j in range(8):
    rl.append("v_mov_b32 %s, %s" % (R(j), E(j)))
redc_valu = sum(1 for l in rl if l.startswith("v_"))

n_valu = sum(1 for l in lines if l.startswith("v_"))
n_mad = sum(1 for l in lines if l.startswith("v_mad") or l.startswith("v_mul"))
with open(OUT, "w") as f:
    f.write("// generated by tools/gen_fr_mul_asm.py %d -- do not edit\n" % TB)
    f.write("// %d VALU instructions (%d multiplies), %d s_nop\n" % (n_valu, n_mad, len(lines) - n_valu))
    f.write("#define ZKIR_FR_MUL_TB %d\n" % TB)
    f.write("#define ZKIR_FR_MUL_VALU %d\n" % n_valu)
    f.write("#define ZKIR_FR_MUL_ASM \\\n")
    for l in lines:
        f.write('  "%s\\n" \\\n' % l)
    f.write("\n")
    f.write("#define ZKIR_FR_REDC_VALU %d\n" % redc_valu)
    f.write("#define ZKIR_FR_REDC_ASM \\\n")
    for l in rl:
        f.write('  "%s\\n" \\\n' % l)
    f.write("\n")
    clob = ", ".join('"v%d"' % (TB + k) for k in range(34))
    f.write("#define ZKIR_FR_MUL_CLOBBERS %s, \"vcc\"\n" % clob)
print("wrote", OUT, n_valu, "VALU", n_mad, "mul")
