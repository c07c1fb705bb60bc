#!/usr/bin/env python3
"""Emit zkir_amd/csrc/hip/interp29_core_asm.inc: the gfx950 instruction stream that interprets
one tape for one wavefront (64 input vectors), for tapes made of + - * and data movement, with field
elements held as 9 limbs of 29 bits (tools/fr29.py says why: no carry-writing add in a product).

Same contract with the C++ around it as the 8 x 32-bit stream of gen_interp_asm.py: one asm statement,
operands read-only until the last two instructions, every register written is in the clobber list, every
memory access goes through a buffer descriptor with a byte count or to LDS.

Values.  Everything the stream keeps (accumulator, slots) is "normal": limbs 0..7 below 2^29, value below
2p, Montgomery form with R = 2^261.  A product of two such values comes out of the multiplier below
1.03 p with normal limbs.  Sums and differences are formed limb by limb (a + b, a + D - b with D a multiple
of p whose limbs dominate b's) and brought back below 2p by one quotient estimate from the top limb
(q = floor(x_8 * 1354 / 2^32) is floor(x / p) or one less), q*p taken from a 32-entry table in LDS, a
limb-wise subtraction and one signed carry sweep.  A CHECK forms t + D - c the same way: when t = c (mod p)
the sweep leaves exactly p (the estimate is always one short on an exact multiple of p), so the test is a
limb compare with p.  OUTPUT leaves Montgomery form with the reduction alone, packs to 8 x 32 bits and
subtracts p once if needed; INPUT unpacks and multiplies by R^2.  Constants arrive in this form already
(the runtime converts the pool at upload: 12 words per constant, 9 used).

Slot layout (both LDS window and device slots): slot s occupies 9 216 bytes;
  byte(s, wave, plane, lane) = s*9216 + wave*2304 + {0, 1024, 2048}[plane] + lane*{16, 16, 4}[plane]
plane 0 holds limbs 0..3, plane 1 limbs 4..7, plane 2 limb 8.  Exchange slots of strand tapes are 2 304
bytes: byte(x, plane, lane) = xbase + x*2304 + ... (same planes).  Operand tags, the victim row (9 216
bytes below the window), BARRIER and the constraint numbers of strand tapes work as in gen_interp_asm.py.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fr29  # noqa: E402

RSRC_WORD3 = 0x00020000  # raw buffer, 32-bit data format
SLOT_BYTES, WAVE_BYTES = 9216, 2304
Q_CONST = 1354  # floor(2^264 / p): floor(x_8 * Q_CONST / 2^32) is floor(x / p) or one less for x below 4000 p
P32 = [(fr29.P >> (32 * j)) & 0xFFFFFFFF for j in range(8)]
R2_29 = fr29.limbs29(fr29.R2)
# this stream holds a tape value x as x * 2^261 mod p.  The input times 2^-256: the input's words times 2^5, i.e. a product
# with 2^266 (the product divides by 2^261); the output times 2^256: the held value times 2^256 / 2^261, i.e. a product
# with the plain integer 2^256
IN_SCALED_29 = fr29.limbs29(pow(2, 266, fr29.P))
OUT_SCALED_29 = fr29.limbs29(1 << 256)


def dominating_multiple(limb_bound, top_bound):
    """limbs of k*p with limbs 0..7 at least limb_bound and limb 8 at least top_bound (units lent downwards)."""
    k = 1
    while True:
        n = fr29.limbs29(k * fr29.P)
        c = [n[0] + (1 << 29)] + [n[j] + (1 << 29) - 1 for j in range(1, 8)] + [n[8] - 1]
        if c[8] >= top_bound and all(v >= limb_bound for v in c[:8]):
            assert fr29.value29(c) == k * fr29.P
            return k, c
        k += 1


# subtrahends are normal and below 2p: limbs below 2^29, top limb below 2^24
D_K, D_LIMBS = dominating_multiple(1 << 29, 1 << 24)

# ---- register plan ---------------------------------------------------------
VB = 24  # first vector register of the stream; register tuples are even-aligned
ACCN = VB            # the multiplier's 64-bit column accumulator
VT0, VT1 = "v%d" % (VB + 2), "v%d" % (VB + 3)


def vreg9(base):
    return ["v%d" % (base + j) for j in range(9)]


def quad(base, half):
    return "v[%d:%d]" % (base + 4 * half, base + 4 * half + 3)


VA, VBB, VACC, VM, VP, VW = VB + 4, VB + 14, VB + 24, VB + 34, VB + 44, VB + 54
A9, B9, ACC9, M9, P9 = vreg9(VA), vreg9(VBB), vreg9(VACC), vreg9(VM), vreg9(VP)
W8 = ["v%d" % (VW + j) for j in range(8)]
AD_A, AD_B, AD_C, AD_4, FIRST = ["v%d" % (VB + 62 + k) for k in range(5)]
V_LAST = VB + 66

SB = 44
D_OPS, D_CST, D_GS, D_IO = ["s[%d:%d]" % (SB + 4 * k, SB + 4 * k + 3) for k in range(4)]
d_word = lambda k, w: "s%d" % (SB + 4 * k + w)
OP = ["s%d" % (SB + 16 + k) for k in range(4)]  # code, dst, a, b
NX = "s[%d:%d]" % (SB + 20, SB + 23)
OP_LO, OP_HI = "s[%d:%d]" % (SB + 16, SB + 17), "s[%d:%d]" % (SB + 18, SB + 19)
NX_LO, NX_HI = "s[%d:%d]" % (SB + 20, SB + 21), "s[%d:%d]" % (SB + 22, SB + 23)
REM, PCOFF = "s%d" % (SB + 24), "s%d" % (SB + 25)
UNSAT, UNSAT_LO, UNSAT_HI = "s[%d:%d]" % (SB + 26, SB + 27), "s%d" % (SB + 26), "s%d" % (SB + 27)
KCHK, CODE = "s%d" % (SB + 28), "s%d" % (SB + 29)
CS = "s[%d:%d]" % (SB + 30, SB + 31)   # carry-out of the multiply-adds (never read)
CC2 = "s[%d:%d]" % (SB + 32, SB + 33)  # a saved borrow / the failing lanes of a check
T0, T1, T2, T3 = ["s%d" % (SB + 34 + k) for k in range(4)]
SNP0, AACC, VMF = "s%d" % (SB + 38), "s%d" % (SB + 39), "s%d" % (SB + 40)
TAG_A, TAG_B = "s%d" % (SB + 42), "s%d" % (SB + 43)  # operand words whose values sit in the a / b registers
CST8 = "s[%d:%d]" % (SB + 44, SB + 51)
CST_LAST = "s%d" % (SB + 52)  # the ninth limb of a constant
cst_word = lambda j: "s%d" % (SB + 44 + j)
TAG_V = "s%d" % (SB + 41)  # device-slot operand word whose value sits in the victim row of LDS
SQ = "s%d" % (SB + 53)    # Q_CONST
NOW = "s[%d:%d]" % (SB + 54, SB + 55)  # the constant-rate clock, read at a barrier (strand tapes with priority turns)
NOW_LO = "s%d" % (SB + 54)
S_LAST = SB + 55
assert S_LAST <= 99  # s100 and above are reserved by the compiler on this target

K_GSLOT, K_LSLOT, K_CONST, K_INPUT, K_ACC, K_NONE, K_XSLOT = 0, 1, 2, 3, 4, 5, 6
OP_INPUT, OP_OUTPUT, OP_COPY, OP_NEG, OP_BARRIER, OP_ADD, OP_SUB, OP_MUL, OP_CHECK = 1, 2, 3, 4, 7, 16, 17, 18, 48
OP_INPUT_SCALED, OP_OUTPUT_SCALED = 8, 9  # the input times 2^-256 / the output times 2^256

# operands of the asm statement (all inputs are read-only)
IN_OPS, IN_NOPS, IN_CST, IN_NCST = "%[ops]", "%[n_ops]", "%[consts]", "%[n_consts]"
IN_GS, IN_GSBYTES, IN_LDSB = "%[gslots]", "%[gslot_bytes]", "%[lds_base]"
IN_INB, IN_OUTB = "%[in_base]", "%[out_base]"
IN_N4, IN_LIVE = "%[n4]", "%[live]"
IN_VLO, IN_VVEC = "%[v_lo]", "%[v_vec4]"
IN_EVICT = "%[evict_below]"  # 0x20000000 when the launch has a victim row, 0 when it has none
IN_XBASE, IN_VX = "%[xbase]", "%[v_x]"  # strand tapes: LDS address of the exchange slots, lane * 16
IN_VLO4, IN_VX4 = "%[v_lo4]", "%[v_x4]"  # plane 2 of a slot: wave * 2304 + 2048 + lane * 4; 2048 + lane * 4
IN_TAB = "%[lds_tab]"  # LDS address of the table of q * p, 48 bytes per entry, q < 32
IN_ONE = "%[one_word]"  # the operand word of the tape's constant 1 (0xffffffff when its pool holds none)
IN_ROT = "%[rot]"      # which of the workgroups resident on its CU this one is (0..3): the phase of its priority
                       # turns; -1: no turns
IN_ROTSH = "%[rot_shift]"  # a turn lasts 1 << rot_shift ticks of the 100 MHz clock
OUT_UNSAT, OUT_FIRST = "%[unsat]", "%[first]"


class Asm:
    def __init__(self):
        self.lines = []
        self.n_label = 0

    def e(self, s):
        self.lines.append(s)

    def label(self, name):
        self.lines.append(name + ":")

    def new(self, stem):
        self.n_label += 1
        return ".Lz29_%s_%d" % (stem, self.n_label)


# ---- arithmetic streams -----------------------------------------------------
def mul_stream(a, x, y, r):
    """r = x*y*2^-261 mod p, normal limbs, below 1.03 p for x, y below 2p; r may alias x or y."""
    fr29.mul29(a.lines, x, y, r, M9, ACCN, P9, SNP0, CS)


def redc_stream(a, x):
    """M9 = x * 2^-261 mod p, normal limbs, below 1.02 p."""
    fr29.mul29(a.lines, x, None, M9, M9, ACCN, P9, SNP0, CS, redc_only=True)


def reduce_stream(a, x):
    """x (9 registers, limbs below 2^32, value below 8p) -> x - q*p with normal limbs, in [0, 2p); exactly p when x
    is a positive multiple of p.  Uses M9 (the table entry) and VT0."""
    a.e("v_mul_hi_u32 %s, %s, %s" % (VT0, x[8], SQ))
    a.e("v_mad_u32_u24 %s, %s, 48, %s" % (VT0, VT0, IN_TAB))
    a.e("ds_read_b128 %s, %s" % (quad(VM, 0), VT0))
    a.e("ds_read_b128 %s, %s offset:16" % (quad(VM, 1), VT0))
    a.e("ds_read_b32 %s, %s offset:32" % (M9[8], VT0))
    a.e("s_waitcnt lgkmcnt(0)")
    for j in range(9):  # may dip below zero limb by limb: the sweep below carries the sign
        a.e("v_subrev_u32 %s, %s, %s" % (x[j], M9[j], x[j]))
    for j in range(8):
        a.e("v_ashrrev_i32 %s, 29, %s" % (VT0, x[j]))
        a.e("v_and_b32 %s, 0x%x, %s" % (x[j], fr29.MASK, x[j]))
        a.e("v_add_u32 %s, %s, %s" % (x[j + 1], x[j + 1], VT0))


def add_stream(a, x, y, r):
    """r = x + y, reduced."""
    fr29.add29(a.lines, r, x, y)
    reduce_stream(a, r)


def sub_stream(a, x, y, r):
    """r = x - y + D, reduced; r may alias x or y."""
    fr29.sub29(a.lines, r, x, y, D_LIMBS)
    reduce_stream(a, r)


# ---- operand access ----------------------------------------------------------
def lds_rw(a, base, addr, addr4, write):
    """the three LDS accesses of one value: planes 0 and 1 through `addr`, plane 2 through `addr4`"""
    regs = vreg9(base)
    if write:
        a.e("ds_write_b128 %s, %s" % (addr, quad(base, 0)))
        a.e("ds_write_b128 %s, %s offset:1024" % (addr, quad(base, 1)))
        a.e("ds_write_b32 %s, %s" % (addr4, regs[8]))
    else:
        a.e("ds_read_b128 %s, %s" % (quad(base, 0), addr))
        a.e("ds_read_b128 %s, %s offset:1024" % (quad(base, 1), addr))
        a.e("ds_read_b32 %s, %s" % (regs[8], addr4))


def fetch(a, sop, dst_base, addr, acc_label=None, stem="f", tag=None):
    """Issue the loads of the operand word in SGPR `sop` into v[dst_base..+8].
    kind ACC: jump to acc_label if given (nothing loaded), else copy the accumulator.
    Device-slot loads set VMF so the caller knows to wait on vmcnt.
    tag: the SGPR that remembers which operand word these registers hold; when it equals
    `sop` the value is already there (no op stored to that slot since: stores reset the tag)
    and nothing is loaded.  AD_4 is the second address register of every access."""
    dst = vreg9(dst_base)
    l_lds, l_glb, l_cst, l_done = a.new(stem + "lds"), a.new(stem + "glb"), a.new(stem + "cst"), a.new(stem + "done")
    l_vhit, l_noev, l_x = a.new(stem + "vhit"), a.new(stem + "noev"), a.new(stem + "x")
    if tag:
        a.e("s_cmp_eq_u32 %s, %s" % (sop, tag))
        a.e("s_cbranch_scc1 %s" % l_done)
        a.e("s_cmp_eq_u32 %s, %s" % (sop, TAG_V))  # TAG_V only ever holds device-slot words
        a.e("s_cbranch_scc1 %s" % l_vhit)
    a.e("s_lshr_b32 %s, %s, 29" % (T0, sop))
    a.e("s_and_b32 %s, %s, 0x1fffffff" % (T1, sop))
    a.e("s_cmp_eq_u32 %s, %d" % (T0, K_ACC))
    if acc_label:
        a.e("s_cbranch_scc1 %s" % acc_label)
    else:
        l_acc = a.new(stem + "acc")
        a.e("s_cbranch_scc1 %s" % l_acc)
    if tag:
        # The registers are about to be overwritten.  If they hold a device-slot value, park it
        # in this wave's victim row of LDS (the 9 216 bytes below the window): the next read of that
        # slot then comes from LDS instead of device memory.
        a.e("s_cmp_lt_u32 %s, %s" % (tag, IN_EVICT))  # kind GSLOT is 0: its words are below 2^29
        a.e("s_cbranch_scc0 %s" % l_noev)
        a.e("s_sub_u32 %s, %s, %d" % (T2, IN_LDSB, SLOT_BYTES))
        a.e("v_add_u32 %s, %s, %s" % (VT1, T2, IN_VLO))
        a.e("v_add_u32 %s, %s, %s" % (AD_4, T2, IN_VLO4))
        lds_rw(a, dst_base, VT1, AD_4, True)
        a.e("s_mov_b32 %s, %s" % (TAG_V, tag))
        a.label(l_noev)
    a.e("s_cmp_eq_u32 %s, %d" % (T0, K_LSLOT))
    a.e("s_cbranch_scc1 %s" % l_lds)
    a.e("s_cmp_eq_u32 %s, %d" % (T0, K_GSLOT))
    a.e("s_cbranch_scc1 %s" % l_glb)
    a.e("s_cmp_eq_u32 %s, %d" % (T0, K_CONST))
    a.e("s_cbranch_scc1 %s" % l_cst)
    a.e("s_cmp_eq_u32 %s, %d" % (T0, K_XSLOT))
    a.e("s_cbranch_scc1 %s" % l_x)
    # any other kind reads as zero (the host validates tapes; this keeps the stream total)
    for j in range(9):
        a.e("v_mov_b32 %s, 0" % dst[j])
    if tag:
        a.e("s_mov_b32 %s, -1" % tag)
    a.e("s_branch %s" % l_done)
    if not acc_label:
        a.label(l_acc)
        if dst_base != VACC:
            for j in range(9):
                a.e("v_mov_b32 %s, %s" % (dst[j], ACC9[j]))
            if tag:
                a.e("s_mov_b32 %s, -1" % tag)
        a.e("s_branch %s" % l_done)
    a.label(l_x)  # exchange slot: another wavefront may rewrite it, so the tag forgets it
    if tag:
        a.e("s_mov_b32 %s, -1" % tag)
    a.e("s_mul_i32 %s, %s, %d" % (T1, T1, WAVE_BYTES))
    a.e("s_add_u32 %s, %s, %s" % (T1, T1, IN_XBASE))
    a.e("v_add_u32 %s, %s, %s" % (addr, T1, IN_VX))
    a.e("v_add_u32 %s, %s, %s" % (AD_4, T1, IN_VX4))
    lds_rw(a, dst_base, addr, AD_4, False)
    a.e("s_branch %s" % l_done)
    a.label(l_cst)
    if tag:
        a.e("s_mov_b32 %s, %s" % (tag, sop))
    a.e("s_mul_i32 %s, %s, 48" % (T1, T1))
    a.e("s_buffer_load_dwordx8 %s, %s, %s" % (CST8, D_CST, T1))
    a.e("s_add_u32 %s, %s, 32" % (T1, T1))
    a.e("s_buffer_load_dword %s, %s, %s" % (CST_LAST, D_CST, T1))
    a.e("s_waitcnt lgkmcnt(0)")
    for j in range(9):
        a.e("v_mov_b32 %s, %s" % (dst[j], cst_word(j)))
    a.e("s_branch %s" % l_done)
    a.label(l_glb)
    if tag:
        a.e("s_mov_b32 %s, %s" % (tag, sop))
    a.e("s_mul_i32 %s, %s, %d" % (T1, T1, SLOT_BYTES))
    a.e("v_add_u32 %s, %s, %s" % (addr, T1, IN_VLO))
    a.e("v_add_u32 %s, %s, %s" % (AD_4, T1, IN_VLO4))
    a.e("buffer_load_dwordx4 %s, %s, %s, 0 offen" % (quad(dst_base, 0), addr, D_GS))
    a.e("buffer_load_dwordx4 %s, %s, %s, 0 offen offset:1024" % (quad(dst_base, 1), addr, D_GS))
    a.e("buffer_load_dword %s, %s, %s, 0 offen" % (dst[8], AD_4, D_GS))
    a.e("s_mov_b32 %s, 1" % VMF)
    a.e("s_branch %s" % l_done)
    a.label(l_lds)
    if tag:
        a.e("s_mov_b32 %s, %s" % (tag, sop))
    a.e("s_mul_i32 %s, %s, %d" % (T1, T1, SLOT_BYTES))
    a.e("s_add_u32 %s, %s, %s" % (T1, T1, IN_LDSB))
    a.e("v_add_u32 %s, %s, %s" % (addr, T1, IN_VLO))
    a.e("v_add_u32 %s, %s, %s" % (AD_4, T1, IN_VLO4))
    lds_rw(a, dst_base, addr, AD_4, False)
    if tag:
        a.e("s_branch %s" % l_done)
        a.label(l_vhit)  # the value waits in the victim row (what the registers held is dropped)
        a.e("s_mov_b32 %s, %s" % (tag, sop))
        a.e("s_sub_u32 %s, %s, %d" % (T2, IN_LDSB, SLOT_BYTES))
        a.e("v_add_u32 %s, %s, %s" % (addr, T2, IN_VLO))
        a.e("v_add_u32 %s, %s, %s" % (AD_4, T2, IN_VLO4))
        lds_rw(a, dst_base, addr, AD_4, False)
    a.label(l_done)


def wait_loads(a):
    """Wait for this op's operand loads: LDS and scalar always, vector memory only if a
    device-slot load was issued (stores of earlier ops stay in flight otherwise)."""
    skip = a.new("wskip")
    a.e("s_waitcnt lgkmcnt(0)")
    a.e("s_cmp_eq_u32 %s, 0" % VMF)
    a.e("s_cbranch_scc1 %s" % skip)
    a.e("s_waitcnt vmcnt(0)")
    a.label(skip)


def build():
    a = Asm()
    # ---- prologue ---------------------------------------------------------
    def descriptor(k, base, nbytes_expr):
        a.e("s_mov_b64 s[%d:%d], %s" % (SB + 4 * k, SB + 4 * k + 1, base))
        a.e("s_and_b32 %s, %s, 0xffff" % (d_word(k, 1), d_word(k, 1)))
        nbytes_expr(d_word(k, 2))
        a.e("s_mov_b32 %s, 0x%08x" % (d_word(k, 3), RSRC_WORD3))
    descriptor(0, IN_OPS, lambda w: a.e("s_lshl_b32 %s, %s, 4" % (w, IN_NOPS)))
    descriptor(1, IN_CST, lambda w: a.e("s_mul_i32 %s, %s, 48" % (w, IN_NCST)))
    descriptor(2, IN_GS, lambda w: a.e("s_mov_b32 %s, %s" % (w, IN_GSBYTES)))
    for j in range(9):
        a.e("v_mov_b32 %s, 0x%08x" % (P9[j], fr29.P_LIMBS[j]))
    for j in range(9):
        a.e("v_mov_b32 %s, 0" % ACC9[j])
    a.e("v_mov_b32 %s, -1" % FIRST)
    a.e("s_mov_b32 %s, -1" % TAG_A)
    a.e("s_mov_b32 %s, -1" % TAG_B)
    a.e("s_mov_b32 %s, -1" % TAG_V)
    a.e("s_mov_b64 %s, 0" % UNSAT)
    a.e("s_mov_b32 %s, 0" % KCHK)
    a.e("s_mov_b32 %s, 0x%08x" % (SNP0, fr29.NP))
    a.e("s_mov_b32 %s, %d" % (SQ, Q_CONST))
    a.e("s_mov_b32 %s, %s" % (REM, IN_NOPS))
    a.e("s_cmp_eq_u32 %s, 0" % REM)
    a.e("s_cbranch_scc1 .Lz29_end")
    a.e("s_buffer_load_dwordx4 %s, %s, 0" % (NX, D_OPS))
    a.e("s_mov_b32 %s, 16" % PCOFF)

    # ---- per-op loop --------------------------------------------------------
    a.label(".Lz29_top")
    a.e("s_waitcnt lgkmcnt(0)")
    a.e("s_mov_b64 %s, %s" % (OP_LO, NX_LO))
    a.e("s_mov_b64 %s, %s" % (OP_HI, NX_HI))
    a.e("s_sub_u32 %s, %s, 1" % (REM, REM))
    a.e("s_cmp_eq_u32 %s, 0" % REM)
    a.e("s_cbranch_scc1 .Lz29_nopf")
    a.e("s_buffer_load_dwordx4 %s, %s, %s" % (NX, D_OPS, PCOFF))
    a.e("s_add_u32 %s, %s, 16" % (PCOFF, PCOFF))
    a.label(".Lz29_nopf")
    a.e("s_and_b32 %s, %s, 0xff" % (CODE, OP[0]))
    a.e("s_mov_b32 %s, 0" % VMF)
    a.e("s_cmp_lt_u32 %s, 16" % CODE)
    a.e("s_cbranch_scc1 .Lz29_unary")

    # A CHECK whose right factor is the constant 1 (the constraint of a signal with a linear value, `value * 1 ==
    # signal`) multiplies nothing: the left factor goes into the comparison as it is (x * 1 = x mod p, and the
    # comparison is one modulo p).
    a.e("s_cmp_eq_u32 %s, %s" % (OP[3], IN_ONE))
    a.e("s_cbranch_scc1 .Lz29_b_is_one")  # rare: the common path falls through
    a.label(".Lz29_b_not_one")
    # binary ops and CHECK: fetch b, then a (or note that a is the accumulator)
    fetch(a, OP[3], VBB, AD_B, stem="b", tag=TAG_B)
    a.e("s_mov_b32 %s, 0" % AACC)
    fetch(a, OP[2], VA, AD_A, acc_label=".Lz29_a_is_acc", stem="a", tag=TAG_A)
    a.e("s_branch .Lz29_have_ab")
    a.label(".Lz29_a_is_acc")
    a.e("s_mov_b32 %s, 1" % AACC)
    a.label(".Lz29_have_ab")
    wait_loads(a)
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_MUL))
    a.e("s_cbranch_scc1 .Lz29_mul")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_ADD))
    a.e("s_cbranch_scc1 .Lz29_add")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_SUB))
    a.e("s_cbranch_scc1 .Lz29_sub")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_CHECK))
    a.e("s_cbranch_scc1 .Lz29_check")
    a.e("s_branch .Lz29_next")  # an opcode this stream does not know: no effect

    a.label(".Lz29_mul")
    a.e("s_cmp_eq_u32 %s, 1" % AACC)
    a.e("s_cbranch_scc1 .Lz29_mul_acc")
    a.label(".Lz29_mul_ab")
    mul_stream(a, A9, B9, ACC9)
    a.e("s_branch .Lz29_store")
    a.label(".Lz29_mul_acc")
    mul_stream(a, ACC9, B9, ACC9)
    a.e("s_branch .Lz29_store")

    # sums and differences are formed limb by limb into the accumulator, then share one reduction
    a.label(".Lz29_add")
    a.e("s_cmp_eq_u32 %s, 1" % AACC)
    a.e("s_cbranch_scc1 .Lz29_add_acc")
    fr29.add29(a.lines, ACC9, A9, B9)
    a.e("s_branch .Lz29_reduce")
    a.label(".Lz29_add_acc")
    fr29.add29(a.lines, ACC9, ACC9, B9)
    a.e("s_branch .Lz29_reduce")

    a.label(".Lz29_sub")
    a.e("s_cmp_eq_u32 %s, 1" % AACC)
    a.e("s_cbranch_scc1 .Lz29_sub_acc")
    a.label(".Lz29_sub_ab")
    fr29.sub29(a.lines, ACC9, A9, B9, D_LIMBS)
    a.e("s_branch .Lz29_reduce")
    a.label(".Lz29_sub_acc")
    fr29.sub29(a.lines, ACC9, ACC9, B9, D_LIMBS)
    a.label(".Lz29_reduce")
    reduce_stream(a, ACC9)
    a.e("s_branch .Lz29_store")

    # CHECK: product into the a registers, then t + D - c, reduced: exactly p when the constraint holds
    a.label(".Lz29_check")
    a.e("s_cmp_eq_u32 %s, 1" % AACC)
    a.e("s_cbranch_scc1 .Lz29_check_acc")
    mul_stream(a, A9, B9, A9)
    a.e("s_branch .Lz29_check_cmp")
    a.label(".Lz29_check_acc")
    mul_stream(a, ACC9, B9, A9)
    a.e("s_branch .Lz29_check_cmp")
    a.label(".Lz29_b_is_one")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_CHECK))
    a.e("s_cbranch_scc0 .Lz29_b_not_one")
    a.label(".Lz29_check_one")  # the "product" is the left factor: into the a registers as any operand would be
    fetch(a, OP[2], VA, AD_A, acc_label=".Lz29_one_acc", stem="p", tag=TAG_A)
    wait_loads(a)
    a.e("s_branch .Lz29_check_cmp")
    a.label(".Lz29_one_acc")
    for j in range(9):
        a.e("v_mov_b32 %s, %s" % (A9[j], ACC9[j]))
    a.label(".Lz29_check_cmp")
    a.e("s_mov_b32 %s, -1" % TAG_A)  # the a registers now hold the product (the comparison below works in place)
    a.e("s_mov_b32 %s, 0" % VMF)
    fetch(a, OP[1], VBB, AD_C, acc_label=".Lz29_cmp_acc", stem="c", tag=TAG_B)
    wait_loads(a)
    fr29.sub29(a.lines, A9, A9, B9, D_LIMBS)
    a.e("s_branch .Lz29_cmp_red")
    a.label(".Lz29_cmp_acc")
    fr29.sub29(a.lines, A9, A9, ACC9, D_LIMBS)
    a.label(".Lz29_cmp_red")
    reduce_stream(a, A9)
    a.e("v_xor_b32 %s, 0x%x, %s" % (VT0, fr29.P_LIMBS[0], A9[0]))
    for j in range(1, 9):
        a.e("v_xor_b32 %s, 0x%x, %s" % (VT1, fr29.P_LIMBS[j], A9[j]))
        a.e("v_or_b32 %s, %s, %s" % (VT0, VT0, VT1))
    a.e("v_cmp_ne_u32 vcc, 0, %s" % VT0)
    a.e("s_nop 1")
    a.e("s_and_b64 %s, vcc, %s" % (CC2, IN_LIVE))
    a.e("s_cbranch_scc0 .Lz29_cmp_ok")
    a.e("s_bcnt1_i32_b64 %s, %s" % (T0, CC2))
    a.e("s_add_u32 %s, %s, %s" % (UNSAT_LO, UNSAT_LO, T0))
    a.e("s_addc_u32 %s, %s, 0" % (UNSAT_HI, UNSAT_HI))
    # the constraint number: carried by the op (number + 1 in bits 8..31) or, when that is 0, the count so far
    a.e("s_lshr_b32 %s, %s, 8" % (T1, OP[0]))
    a.e("s_cbranch_scc1 .Lz29_cmp_id")
    a.e("s_add_u32 %s, %s, 1" % (T1, KCHK))
    a.label(".Lz29_cmp_id")
    a.e("s_sub_u32 %s, %s, 1" % (T1, T1))
    a.e("v_mov_b32 %s, %s" % (VT0, T1))
    a.e("v_mov_b32 %s, -1" % VT1)
    a.e("v_cndmask_b32_e64 %s, %s, %s, %s" % (VT0, VT1, VT0, CC2))
    a.e("v_min_u32 %s, %s, %s" % (FIRST, FIRST, VT0))
    a.label(".Lz29_cmp_ok")
    a.e("s_add_u32 %s, %s, 1" % (KCHK, KCHK))
    a.e("s_branch .Lz29_next")

    # ---- INPUT, OUTPUT, COPY, NEG ------------------------------------------
    a.label(".Lz29_unary")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_OUTPUT))
    a.e("s_cbranch_scc1 .Lz29_output")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_OUTPUT_SCALED))
    a.e("s_cbranch_scc1 .Lz29_output")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_INPUT))
    a.e("s_cbranch_scc1 .Lz29_input")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_INPUT_SCALED))
    a.e("s_cbranch_scc1 .Lz29_input")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_COPY))
    a.e("s_cbranch_scc1 .Lz29_copy")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_NEG))
    a.e("s_cbranch_scc1 .Lz29_neg")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_BARRIER))
    a.e("s_cbranch_scc0 .Lz29_next")  # NOP and anything unknown
    # BARRIER: everything this wavefront stored is visible to the workgroup once its memory
    # operations have completed; then wait for the other strands
    a.e("s_cmp_eq_u32 %s, -1" % IN_ROT)
    a.e("s_cbranch_scc0 .Lz29_barrier_turn")
    a.e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    a.e("s_barrier")
    a.e("s_branch .Lz29_next")
    # The SIMD issues from its oldest wavefront first, so of the workgroups resident on a CU the first one placed
    # runs ahead and the last one finishes alone, on SIMDs it cannot fill (measured on four workgroups of four
    # strands with one work item each: 56, 69, 87 and 106 ms for the same work).  Priorities that turn with the
    # clock, each resident workgroup in a different phase, give every workgroup the same share: they finish
    # together.  (Turns counted in barriers instead of clock ticks drift apart between workgroups, which then hold
    # equal priorities much of the time: a third of the gain.  Turns shorter than 40 us or longer than 1 ms lose too.)
    a.label(".Lz29_barrier_turn")
    a.e("s_memrealtime %s" % NOW)
    a.e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    a.e("s_barrier")
    a.e("s_lshr_b32 %s, %s, %s" % (T0, NOW_LO, IN_ROTSH))
    a.e("s_add_u32 %s, %s, %s" % (T0, T0, IN_ROT))
    a.e("s_and_b32 %s, %s, 3" % (T0, T0))
    a.e("s_cmp_lt_u32 %s, 2" % T0)
    a.e("s_cbranch_scc1 .Lz29_prio01")
    a.e("s_cmp_eq_u32 %s, 2" % T0)
    a.e("s_cbranch_scc1 .Lz29_prio2")
    a.e("s_setprio 3")
    a.e("s_branch .Lz29_next")
    a.label(".Lz29_prio2")
    a.e("s_setprio 2")
    a.e("s_branch .Lz29_next")
    a.label(".Lz29_prio01")
    a.e("s_cmp_eq_u32 %s, 0" % T0)
    a.e("s_cbranch_scc1 .Lz29_prio0")
    a.e("s_setprio 1")
    a.e("s_branch .Lz29_next")
    a.label(".Lz29_prio0")
    a.e("s_setprio 0")
    a.e("s_branch .Lz29_next")

    def io_descriptor(base, index_reg):
        # element `index` of the tape's inputs (outputs): 8 limb rows of n vectors, 32*n bytes
        a.e("s_lshl_b32 %s, %s, 3" % (d_word(3, 2), IN_N4))  # 32*n bytes: the descriptor's byte count
        a.e("s_mul_i32 %s, %s, %s" % (T2, index_reg, d_word(3, 2)))
        a.e("s_mul_hi_u32 %s, %s, %s" % (T3, index_reg, d_word(3, 2)))
        a.e("s_mov_b64 s[%d:%d], %s" % (SB + 12, SB + 13, base))
        a.e("s_add_u32 %s, %s, %s" % (d_word(3, 0), d_word(3, 0), T2))
        a.e("s_addc_u32 %s, %s, %s" % (d_word(3, 1), d_word(3, 1), T3))
        a.e("s_and_b32 %s, %s, 0xffff" % (d_word(3, 1), d_word(3, 1)))
        a.e("s_mov_b32 %s, 0x%08x" % (d_word(3, 3), RSRC_WORD3))

    def limb_offsets():
        # byte offset of limb row j, one scalar register each (none is rewritten while a
        # memory instruction that names it may still be reading it)
        a.e("s_mov_b32 %s, 0" % cst_word(0))
        for j in range(1, 8):
            a.e("s_add_u32 %s, %s, %s" % (cst_word(j), cst_word(j - 1), IN_N4))

    a.label(".Lz29_input")
    a.e("s_and_b32 %s, %s, 0x1fffffff" % (T1, OP[2]))
    io_descriptor(IN_INB, T1)
    limb_offsets()
    for j in range(8):
        a.e("buffer_load_dword %s, %s, %s, %s offen" % (W8[j], IN_VVEC, D_IO, cst_word(j)))
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_INPUT_SCALED))
    a.e("s_cbranch_scc1 .Lz29_input_scaled")
    for j in range(9):
        a.e("v_mov_b32 %s, 0x%08x" % (B9[j], R2_29[j]))
    a.e("s_branch .Lz29_input_mul")
    a.label(".Lz29_input_scaled")
    for j in range(9):
        a.e("v_mov_b32 %s, 0x%08x" % (B9[j], IN_SCALED_29[j]))
    a.label(".Lz29_input_mul")
    a.e("s_mov_b32 %s, -1" % TAG_A)
    a.e("s_mov_b32 %s, -1" % TAG_B)
    a.e("s_waitcnt vmcnt(0)")
    fr29.unpack29(a.lines, A9, W8)
    a.e("s_branch .Lz29_mul_ab")

    a.label(".Lz29_output")
    a.e("s_cmp_eq_u64 %s, 0" % IN_OUTB)  # nobody wants the outputs: skip the conversion too
    a.e("s_cbranch_scc1 .Lz29_next")
    fetch(a, OP[2], VA, AD_A, acc_label=".Lz29_out_acc", stem="o", tag=TAG_A)
    wait_loads(a)
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_OUTPUT_SCALED))
    a.e("s_cbranch_scc1 .Lz29_out_scaled")
    redc_stream(a, A9)
    a.e("s_branch .Lz29_out_store")
    a.label(".Lz29_out_scaled")  # times 2^256 / 2^261: a product with the integer 2^256, below 1.07 p
    a.e("s_mov_b32 %s, -1" % TAG_B)
    for j in range(9):
        a.e("v_mov_b32 %s, 0x%08x" % (B9[j], OUT_SCALED_29[j]))
    mul_stream(a, A9, B9, M9)
    a.e("s_branch .Lz29_out_store")
    a.label(".Lz29_out_acc")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_OUTPUT_SCALED))
    a.e("s_cbranch_scc1 .Lz29_out_acc_scaled")
    redc_stream(a, ACC9)
    a.e("s_branch .Lz29_out_store")
    a.label(".Lz29_out_acc_scaled")
    a.e("s_mov_b32 %s, -1" % TAG_B)
    for j in range(9):
        a.e("v_mov_b32 %s, 0x%08x" % (B9[j], OUT_SCALED_29[j]))
    mul_stream(a, ACC9, B9, M9)
    a.label(".Lz29_out_store")
    # the value is below 1.02 p: pack it to 8 x 32 bits and subtract p once if it is not below p
    fr29.pack29(a.lines, W8, M9, VT0)
    a.e("s_mov_b32 %s, -1" % TAG_B)  # the b registers take p's words and the difference
    for j in range(8):
        a.e("v_mov_b32 %s, 0x%08x" % (B9[j], P32[j]))
    a.e("v_sub_co_u32 %s, vcc, %s, %s" % (B9[0], W8[0], B9[0]))
    for j in range(1, 8):
        a.e("s_nop 0")
        a.e("v_subb_co_u32 %s, vcc, %s, %s, vcc" % (B9[j], W8[j], B9[j]))
    a.e("s_nop 1")
    for j in range(8):  # borrow: the value was below p
        a.e("v_cndmask_b32 %s, %s, %s, vcc" % (W8[j], B9[j], W8[j]))
    io_descriptor(IN_OUTB, OP[1])
    limb_offsets()
    a.e("s_mov_b64 exec, %s" % IN_LIVE)
    for j in range(8):
        a.e("buffer_store_dword %s, %s, %s, %s offen" % (W8[j], IN_VVEC, D_IO, cst_word(j)))
    a.e("s_mov_b64 exec, -1")
    a.e("s_branch .Lz29_next")

    a.label(".Lz29_copy")
    fetch(a, OP[2], VACC, AD_A, stem="k")
    wait_loads(a)
    a.e("s_branch .Lz29_store")

    a.label(".Lz29_neg")  # 0 - a
    fetch(a, OP[2], VBB, AD_B, stem="n", tag=TAG_B)
    a.e("s_mov_b32 %s, -1" % TAG_A)
    for j in range(9):
        a.e("v_mov_b32 %s, 0" % A9[j])
    wait_loads(a)
    a.e("s_branch .Lz29_sub_ab")

    # ---- store the accumulator where the op says ---------------------------
    a.label(".Lz29_store")
    # a value remembered in the operand registers dies when its slot is overwritten
    a.e("s_cmp_eq_u32 %s, %s" % (OP[1], TAG_A))
    a.e("s_cselect_b32 %s, -1, %s" % (TAG_A, TAG_A))
    a.e("s_cmp_eq_u32 %s, %s" % (OP[1], TAG_B))
    a.e("s_cselect_b32 %s, -1, %s" % (TAG_B, TAG_B))
    a.e("s_cmp_eq_u32 %s, %s" % (OP[1], TAG_V))
    a.e("s_cselect_b32 %s, -1, %s" % (TAG_V, TAG_V))
    a.e("s_lshr_b32 %s, %s, 29" % (T0, OP[1]))
    a.e("s_and_b32 %s, %s, 0x1fffffff" % (T1, OP[1]))
    a.e("s_cmp_eq_u32 %s, %d" % (T0, K_XSLOT))
    a.e("s_cbranch_scc1 .Lz29_st_x")
    a.e("s_mul_i32 %s, %s, %d" % (T1, T1, SLOT_BYTES))
    a.e("s_cmp_eq_u32 %s, %d" % (T0, K_LSLOT))
    a.e("s_cbranch_scc1 .Lz29_st_lds")
    a.e("s_cmp_eq_u32 %s, %d" % (T0, K_GSLOT))
    a.e("s_cbranch_scc0 .Lz29_next")
    a.e("v_add_u32 %s, %s, %s" % (AD_C, T1, IN_VLO))
    a.e("v_add_u32 %s, %s, %s" % (AD_4, T1, IN_VLO4))
    a.e("buffer_store_dwordx4 %s, %s, %s, 0 offen" % (quad(VACC, 0), AD_C, D_GS))
    a.e("buffer_store_dwordx4 %s, %s, %s, 0 offen offset:1024" % (quad(VACC, 1), AD_C, D_GS))
    a.e("buffer_store_dword %s, %s, %s, 0 offen" % (ACC9[8], AD_4, D_GS))
    a.e("s_nop 1")
    a.e("s_branch .Lz29_next")
    a.label(".Lz29_st_lds")
    a.e("s_add_u32 %s, %s, %s" % (T1, T1, IN_LDSB))
    a.e("v_add_u32 %s, %s, %s" % (AD_C, T1, IN_VLO))
    a.e("v_add_u32 %s, %s, %s" % (AD_4, T1, IN_VLO4))
    lds_rw(a, VACC, AD_C, AD_4, True)
    a.e("s_branch .Lz29_next")
    a.label(".Lz29_st_x")
    a.e("s_mul_i32 %s, %s, %d" % (T1, T1, WAVE_BYTES))
    a.e("s_add_u32 %s, %s, %s" % (T1, T1, IN_XBASE))
    a.e("v_add_u32 %s, %s, %s" % (AD_C, T1, IN_VX))
    a.e("v_add_u32 %s, %s, %s" % (AD_4, T1, IN_VX4))
    lds_rw(a, VACC, AD_C, AD_4, True)
    a.label(".Lz29_next")
    a.e("s_cmp_lg_u32 %s, 0" % REM)
    a.e("s_cbranch_scc1 .Lz29_top")

    a.label(".Lz29_end")
    a.e("s_setprio 0")
    a.e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    a.e("s_mov_b64 %s, %s" % (OUT_UNSAT, UNSAT))
    a.e("v_mov_b32 %s, %s" % (OUT_FIRST, FIRST))
    return a.lines


def rate_lines():
    """Four Montgomery products per trip in registers, operands rotating through four register sets: the ceiling of
    any tape made of products.  Operands: %[seed] (VGPR), %[iters] (SGPR, at least 1), result %[res] (VGPR)."""
    out = []
    X, Y, Z, W = A9, B9, ACC9, vreg9(VW)
    for j in range(9):
        out.append("v_mov_b32 %s, 0x%08x" % (P9[j], fr29.P_LIMBS[j]))
    out.append("s_mov_b32 %s, 0x%08x" % (SNP0, fr29.NP))
    out.append("s_mov_b32 %s, %%[iters]" % REM)
    for regs, mul in ((X, 0x9E3779B1), (Y, 0x85EBCA77)):
        for j in range(9):
            out.append("v_mul_lo_u32 %s, %%[seed], %s" % (regs[j], P9[j]))
            out.append("v_add_u32 %s, 0x%x, %s" % (regs[j], (mul * (j + 1)) & 0xFFFFFFF, regs[j]))
            out.append("v_and_b32 %s, 0x%x, %s" % (regs[j], fr29.MASK if j < 8 else 0x1FFFFF, regs[j]))
    out.append(".Lz29_rate_top:")
    fr29.mul29(out, X, Y, Z, M9, ACCN, P9, SNP0, CS)
    fr29.mul29(out, Y, Z, W, M9, ACCN, P9, SNP0, CS)
    fr29.mul29(out, Z, W, X, M9, ACCN, P9, SNP0, CS)
    fr29.mul29(out, W, X, Y, M9, ACCN, P9, SNP0, CS)
    out.append("s_sub_u32 %s, %s, 1" % (REM, REM))
    out.append("s_cmp_lg_u32 %s, 0" % REM)
    out.append("s_cbranch_scc1 .Lz29_rate_top")
    out.append("v_mov_b32 %%[res], %s" % X[0])
    for r in X[1:] + Y:
        out.append("v_xor_b32 %%[res], %%[res], %s" % r)
    return out


def clobbers():
    v = ['"v%d"' % k for k in range(VB, V_LAST + 1)]
    s = ['"s%d"' % k for k in range(SB, S_LAST + 1)]
    return ", ".join(v + s + ['"vcc"', '"scc"', '"memory"'])


def render(lines):
    out = ["// generated by tools/gen_interp29_asm.py -- do not edit",
           "// vector registers v%d..v%d, scalar registers s%d..s%d" % (VB, V_LAST, SB, S_LAST),
           "#define ZKIR_INTERP29_CORE_ASM \\"]
    for l in lines:
        out.append('  "%s\\n" \\' % l)
    out.append("")
    out.append("#define ZKIR_INTERP29_CORE_CLOBBERS %s" % clobbers())
    # the field's constants in this representation, for the C++ around the stream
    out.append("#define ZKIR_FR29_P_LIMBS %s" % ", ".join("0x%08xu" % v for v in fr29.P_LIMBS))
    out.append("#define ZKIR_FR29_R2_LIMBS %s" % ", ".join("0x%08xu" % v for v in R2_29))
    out.append("#define ZKIR_FR29_NP 0x%08xu" % fr29.NP)
    out.append("#define ZKIR_INTERP29_CORE_VGPR_LAST %d" % V_LAST)
    out.append("// four products per trip in registers (the rate probe); same clobber list")
    out.append("#define ZKIR_FR29_RATE_ASM \\")
    for l in rate_lines():
        out.append('  "%s\\n" \\' % l)
    out.append("")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "zkir_amd", "csrc", "hip", "interp29_core_asm.inc")
    lines = build()
    with open(path, "w") as f:
        f.write(render(lines))
    n_valu = sum(1 for l in lines if l.startswith("v_"))
    print("wrote %s: %d lines, %d vector ALU instructions" % (path, len(lines), n_valu))
