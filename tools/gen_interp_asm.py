#!/usr/bin/env python3
"""Emit zkir_amd/csrc/hip/interp_core_asm.inc: the gfx950 instruction stream that interprets
one tape for one wavefront (64 input vectors), for tapes made of + - * and data movement.

The stream is one asm statement inside zkir_interp_lean_kernel.  Its contract with the C++
around it: every operand of the statement is read-only until the last two instructions
(which write the two results), every register the stream writes is a physical register
named in the clobber list, every memory access goes through a buffer descriptor with a
byte count (the hardware drops accesses beyond it) or to LDS.

Per op: the 16-byte op arrives by scalar load one op ahead; operands are fetched by kind
(LDS window / device slot / constant pool / accumulator); + - * run on fixed registers,
the product with the Montgomery stream of gen_fr_mul_asm.py; the result stays in the
accumulator registers and is stored if the op names a destination.

Operand registers carry tags: the a and b registers remember which operand word they hold, so
an op that reads what the previous ops already fetched loads nothing; a device-slot value pushed
out of them is parked in one victim row of LDS (the 8 KiB below the window; `lds_base` is the
window's start) and read back from there if it is the next thing wanted (launches whose LDS
window is large run without the row: `evict_below` = 0 switches the parking off).  A store to a slot
clears every tag that names it.

Slot layout (both LDS window and device slots): slot s occupies 8 KiB;
  byte(s, wave, plane, lane) = s*8192 + wave*2048 + plane*1024 + lane*16
plane 0 holds limbs 0..3, plane 1 limbs 4..7, so a wave reads or writes a plane as one
1 KiB row.

Strand tapes (four wavefronts of a workgroup run four op streams on the same 64 vectors): the
stream is the same; a wavefront reads its own strand's ops, keeps its own rows of every slot
(the wave term of the layout above), and meets the others in exchange slots of 2 KiB,
  byte(x, plane, lane) = xbase + x*2048 + plane*1024 + lane*16       (LDS, operand kind 6)
which are never remembered in the operand tags, and at BARRIER ops (all memory operations
waited for, then s_barrier).  A CHECK op of such a tape carries its constraint number + 1 in
bits 8..31 of its code word; 0 there means "count the checks" (plain tapes).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fr29  # noqa: E402

P_LIMBS = [0xf0000001, 0x43e1f593, 0x79b97091, 0x2833e848, 0x8181585d, 0xb85045b6, 0xe131a029, 0x30644e72]
R2_LIMBS = [0xae216da7, 0x1bb8e645, 0xe35c59e3, 0x53fe3ab1, 0x53bb8085, 0x8c49833d, 0x7f4e44a5, 0x0216d0b1]
NP0 = 0xefffffff
RSRC_WORD3 = 0x00020000  # raw buffer, 32-bit data format

# ---- register plan ---------------------------------------------------------
VB = 24  # first vector register of the stream (even: 64-bit operands must be even-aligned)
E = lambda j: "v%d" % (VB + 2 * j) if j < 8 else "v%d" % (VB + 32)
Z = lambda j: "v%d" % (VB + 2 * j + 1)
EZ = lambda j: "v[%d:%d]" % (VB + 2 * j, VB + 2 * j + 1)
L = lambda j: "v%d" % (VB + 16 + 2 * j)
H = lambda j: "v%d" % (VB + 17 + 2 * j)
D = lambda j: "v[%d:%d]" % (VB + 16 + 2 * j, VB + 17 + 2 * j)
M = "v%d" % (VB + 33)


def vreg8(base):
    return ["v%d" % (base + j) for j in range(8)]


def quad(base, half):
    return "v[%d:%d]" % (base + 4 * half, base + 4 * half + 3)


# The product works on 9 limbs of 29 bits (tools/fr29.py) inside the registers the other streams use as scratch:
# the 64-bit column accumulator, one temporary, both operands unpacked, the reduction multipliers (which become
# the result), then the packed result (over the first operand) and its difference with p (over the second).
ACCN, TT = VB, "v%d" % (VB + 2)
UA = ["v%d" % (VB + 3 + j) for j in range(9)]
UB = ["v%d" % (VB + 12 + j) for j in range(9)]
M9 = ["v%d" % (VB + 21 + j) for j in range(9)]
W8, T8 = UA[:8], UB[:8]
assert VB + 29 <= VB + 33

VA, VBB, VACC, VP = VB + 34, VB + 42, VB + 50, VB + 58
A8, B8, ACC8, P8 = vreg8(VA), vreg8(VBB), vreg8(VACC), vreg8(VP)
AD_A, AD_B, AD_C, FIRST, VT0, VT1 = ["v%d" % (VB + 66 + k) for k in range(6)]
V_LAST = VB + 71

S_FIRST = 38
SP29 = ["s%d" % (S_FIRST + j) for j in range(9)]  # p as 9 limbs of 29 bits
SNP29 = "s%d" % (S_FIRST + 9)                      # -1/p mod 2^29
SB = 48
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
cst_word = lambda j: "s%d" % (SB + 44 + j)
TAG_V = "s%d" % (SB + 41)  # device-slot operand word whose value sits in the victim row of LDS
NOW = "s[%d:%d]" % (SB + 36, SB + 37)  # the constant-rate clock, read at a barrier (strand tapes with priority turns):
NOW_LO = "s%d" % (SB + 36)             # the temporaries T2 and T3, which hold nothing across ops
S_LAST = SB + 51
assert S_LAST <= 99  # s100 and above are reserved by the compiler on this target

K_GSLOT, K_LSLOT, K_CONST, K_INPUT, K_ACC, K_NONE, K_XSLOT = 0, 1, 2, 3, 4, 5, 6
OP_INPUT, OP_OUTPUT, OP_COPY, OP_NEG, OP_BARRIER, OP_ADD, OP_SUB, OP_MUL, OP_CHECK = 1, 2, 3, 4, 7, 16, 17, 18, 48
OP_INPUT_SCALED, OP_OUTPUT_SCALED = 8, 9  # the input times 2^-256 / the output times 2^256: the held words as they are

# operands of the asm statement (all inputs are read-only)
IN_OPS, IN_NOPS, IN_CST, IN_NCST = "%[ops]", "%[n_ops]", "%[consts]", "%[n_consts]"
IN_GS, IN_GSBYTES, IN_LDSB = "%[gslots]", "%[gslot_bytes]", "%[lds_base]"
IN_INB, IN_OUTB = "%[in_base]", "%[out_base]"
IN_N4, IN_LIVE = "%[n4]", "%[live]"
IN_VLO, IN_VVEC = "%[v_lo]", "%[v_vec4]"
IN_EVICT = "%[evict_below]"  # 0x20000000 when the launch has a victim row, 0 when it has none
IN_XBASE, IN_VX = "%[xbase]", "%[v_x]"  # strand tapes: LDS address of the exchange slots, lane * 16
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
        return ".Lzk_%s_%d" % (stem, self.n_label)


# ---- arithmetic streams -----------------------------------------------------
def mul_stream(a, x, y, r, after_unpack=None):
    """r = x*y*2^-256 mod p, below p, for x, y below p; x, y, r lists of 8 registers; r may alias x or y, or be W8.
    after_unpack: called once x and y have been read for the last time (the registers of x and y are free from
    there on unless r names them): the place where the next op's operands are sent for.
    The 8 x 32-bit operands are unpacked to 9 limbs of 29 bits, x as 32 x, and multiplied with the radix-2^261
    Montgomery product of tools/fr29.py (162 multiply-adds into 64-bit column sums and no add that writes a carry:
    on gfx950 such an add costs an issue slot like a multiply): 32 x y 2^-261 = x y 2^-256.  The result, below
    1.3 p with limbs below 2^29, is packed back to 8 words and p is subtracted once if it fits."""
    fr29.unpack29(a.lines, UA, x, times32=True)
    fr29.unpack29(a.lines, UB, y)
    if after_unpack:
        after_unpack()
    fr29.mul29(a.lines, UA, UB, M9, M9, ACCN, SP29, SNP29, CS)
    fr29.pack29(a.lines, W8, M9, TT)
    a.e("v_sub_co_u32 %s, vcc, %s, %s" % (T8[0], W8[0], P8[0]))
    for j in range(1, 8):
        a.e("s_nop 0")
        a.e("v_subb_co_u32 %s, vcc, %s, %s, vcc" % (T8[j], W8[j], P8[j]))
    a.e("s_nop 1")
    for j in range(8):  # borrow: the product was below p
        a.e("v_cndmask_b32 %s, %s, %s, vcc" % (r[j], T8[j], W8[j]))


def redc_stream(a, x):
    """E(0..7) = x * 2^-256 mod p for x < p."""
    for j in range(8):  # the high halves of the addend pairs (the product stream uses these registers too)
        a.e("v_mov_b32 %s, 0" % Z(j))
    for j in range(8):
        a.e("v_mov_b32 %s, %s" % (E(j), x[j]))
    a.e("v_mul_lo_u32 %s, %s, %s" % (M, E(0), SNP0))
    for j in range(8):
        a.e("v_mad_u64_u32 %s, %s, %s, %s, %s" % (D(j), CS, M, P8[j], EZ(j)))
    for i in range(8):
        last = i == 7
        a.e("v_add_co_u32 %s, vcc, %s, %s" % (E(0), L(1), H(0)))
        a.e("s_nop 0" if last else "v_mul_lo_u32 %s, %s, %s" % (M, E(0), SNP0))
        for j in range(2, 9):
            if j < 8:
                a.e("v_addc_co_u32 %s, vcc, %s, %s, vcc" % (E(j - 1), L(j), H(j - 1)))
            else:
                a.e("v_addc_co_u32 %s, vcc, 0, %s, vcc" % (E(7), H(7)))
            if not last:
                a.e("v_mad_u64_u32 %s, %s, %s, %s, %s" % (D(j - 2), CS, M, P8[j - 2], EZ(j - 2)))
            elif j < 8:
                a.e("s_nop 0")
        if not last:
            a.e("v_mad_u64_u32 %s, %s, %s, %s, %s" % (D(7), CS, M, P8[7], EZ(7)))


def add_stream(a, x, y, r):
    """r = x + y mod p."""
    a.e("v_add_co_u32 %s, vcc, %s, %s" % (E(0), x[0], y[0]))
    for j in range(1, 8):
        a.e("s_nop 0")
        a.e("v_addc_co_u32 %s, vcc, %s, %s, vcc" % (E(j), x[j], y[j]))
    a.e("s_nop 0")
    a.e("v_sub_co_u32 %s, vcc, %s, %s" % (L(0), E(0), P8[0]))
    for j in range(1, 8):
        a.e("s_nop 0")
        a.e("v_subb_co_u32 %s, vcc, %s, %s, vcc" % (L(j), E(j), P8[j]))
    a.e("s_nop 1")
    for j in range(8):  # borrow: the sum was below p
        a.e("v_cndmask_b32 %s, %s, %s, vcc" % (r[j], L(j), E(j)))


def sub_stream(a, x, y, r):
    """r = x - y mod p."""
    a.e("v_sub_co_u32 %s, vcc, %s, %s" % (E(0), x[0], y[0]))
    for j in range(1, 8):
        a.e("s_nop 0")
        a.e("v_subb_co_u32 %s, vcc, %s, %s, vcc" % (E(j), x[j], y[j]))
    a.e("s_nop 1")
    a.e("s_mov_b64 %s, vcc" % CC2)
    a.e("v_add_co_u32 %s, vcc, %s, %s" % (L(0), E(0), P8[0]))
    for j in range(1, 8):
        a.e("s_nop 0")
        a.e("v_addc_co_u32 %s, vcc, %s, %s, vcc" % (L(j), E(j), P8[j]))
    for j in range(8):  # borrow: take the difference plus p
        a.e("v_cndmask_b32_e64 %s, %s, %s, %s" % (r[j], E(j), L(j), CC2))


# ---- operand access ----------------------------------------------------------
def fetch(a, sop, dst_base, addr, acc_label=None, stem="f", tag=None, vmf=1):
    """Issue the loads of the operand word in SGPR `sop` into v[dst_base..+7].
    kind ACC: jump to acc_label if given (nothing loaded), else copy the accumulator.
    Device-slot loads set VMF so the caller knows to wait on vmcnt (vmf = 1: this op's loads; 2: loads sent for
    ahead of time for the next op, which becomes 1 when that op begins).
    tag: the SGPR that remembers which operand word these registers hold; when it equals
    `sop` the value is already there (no op stored to that slot since: stores reset the tag)
    and nothing is loaded."""
    dst = vreg8(dst_base)
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
        # in this wave's victim row of LDS (the 8 KiB below the window): the next read of that
        # slot then comes from LDS instead of device memory.
        a.e("s_cmp_lt_u32 %s, %s" % (tag, IN_EVICT))  # kind GSLOT is 0: its words are below 2^29
        a.e("s_cbranch_scc0 %s" % l_noev)
        a.e("s_sub_u32 %s, %s, 8192" % (T2, IN_LDSB))
        a.e("v_add_u32 %s, %s, %s" % (VT1, T2, IN_VLO))
        a.e("ds_write_b128 %s, %s" % (VT1, quad(dst_base, 0)))
        a.e("ds_write_b128 %s, %s offset:1024" % (VT1, quad(dst_base, 1)))
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
    for j in range(8):
        a.e("v_mov_b32 %s, 0" % dst[j])
    if tag:
        a.e("s_mov_b32 %s, -1" % tag)
    a.e("s_branch %s" % l_done)
    if not acc_label:
        a.label(l_acc)
        if dst_base != VACC:
            for j in range(8):
                a.e("v_mov_b32 %s, %s" % (dst[j], ACC8[j]))
            if tag:
                a.e("s_mov_b32 %s, -1" % tag)
        a.e("s_branch %s" % l_done)
    a.label(l_x)  # exchange slot: another wavefront may rewrite it, so the tag forgets it
    if tag:
        a.e("s_mov_b32 %s, -1" % tag)
    a.e("s_lshl_b32 %s, %s, 11" % (T1, T1))
    a.e("s_add_u32 %s, %s, %s" % (T1, T1, IN_XBASE))
    a.e("v_add_u32 %s, %s, %s" % (addr, T1, IN_VX))
    a.e("ds_read_b128 %s, %s" % (quad(dst_base, 0), addr))
    a.e("ds_read_b128 %s, %s offset:1024" % (quad(dst_base, 1), addr))
    a.e("s_branch %s" % l_done)
    a.label(l_cst)
    if tag:
        a.e("s_mov_b32 %s, %s" % (tag, sop))
    a.e("s_lshl_b32 %s, %s, 5" % (T1, T1))
    a.e("s_buffer_load_dwordx8 %s, %s, %s" % (CST8, D_CST, T1))
    a.e("s_waitcnt lgkmcnt(0)")
    for j in range(8):
        a.e("v_mov_b32 %s, %s" % (dst[j], cst_word(j)))
    a.e("s_branch %s" % l_done)
    a.label(l_glb)
    if tag:
        a.e("s_mov_b32 %s, %s" % (tag, sop))
    a.e("s_lshl_b32 %s, %s, 13" % (T1, T1))
    a.e("v_add_u32 %s, %s, %s" % (addr, T1, IN_VLO))
    a.e("buffer_load_dwordx4 %s, %s, %s, 0 offen" % (quad(dst_base, 0), addr, D_GS))
    a.e("buffer_load_dwordx4 %s, %s, %s, 0 offen offset:1024" % (quad(dst_base, 1), addr, D_GS))
    a.e("s_mov_b32 %s, %d" % (VMF, vmf))
    a.e("s_branch %s" % l_done)
    a.label(l_lds)
    if tag:
        a.e("s_mov_b32 %s, %s" % (tag, sop))
    a.e("s_lshl_b32 %s, %s, 13" % (T1, T1))
    a.e("s_add_u32 %s, %s, %s" % (T1, T1, IN_LDSB))
    a.e("v_add_u32 %s, %s, %s" % (addr, T1, IN_VLO))
    a.e("ds_read_b128 %s, %s" % (quad(dst_base, 0), addr))
    a.e("ds_read_b128 %s, %s offset:1024" % (quad(dst_base, 1), addr))
    if tag:
        a.e("s_branch %s" % l_done)
        a.label(l_vhit)  # the value waits in the victim row (what the registers held is dropped)
        a.e("s_mov_b32 %s, %s" % (tag, sop))
        a.e("s_sub_u32 %s, %s, 8192" % (T2, IN_LDSB))
        a.e("v_add_u32 %s, %s, %s" % (addr, T2, IN_VLO))
        a.e("ds_read_b128 %s, %s" % (quad(dst_base, 0), addr))
        a.e("ds_read_b128 %s, %s offset:1024" % (quad(dst_base, 1), addr))
    a.label(l_done)


def wait_loads(a):
    """Wait for this op's operand loads: LDS and scalar always, vector memory only if a
    device-slot load was issued (stores of earlier ops stay in flight otherwise)."""
    skip = a.new("wskip")
    a.e("s_waitcnt lgkmcnt(0)")
    a.e("s_and_b32 %s, %s, 1" % (T0, VMF))
    a.e("s_cbranch_scc0 %s" % skip)
    a.e("s_waitcnt vmcnt(0)")
    a.label(skip)


NXW = ["s%d" % (SB + 20 + k) for k in range(4)]  # the next op: code, dst, a, b


def prefetch_next(a, stem, host_is_check):
    """Inside a product, once its operands are unpacked: if the next op is a binary op or a CHECK, send for its b
    and a operands now (same fetch as that op would make, into the same registers, with the same tags and the same
    use of the victim row), so that device-memory latency passes behind the product.  The next op then finds its
    operands by their tags and waits for them (VMF bit 1 becomes bit 0 at its start).
    Not done: for the last op; when the next op reads what this op is about to store; when the next op is a CHECK
    against the constant 1 (it fetches no b); inside a CHECK whose right-hand side is not the accumulator (it is
    fetched into the b registers after the product)."""
    skip, after_b = a.new(stem + "skip"), a.new(stem + "ab")
    a.e("s_waitcnt lgkmcnt(0)")  # the next op has arrived
    a.e("s_cmp_eq_u32 %s, 0" % REM)
    a.e("s_cbranch_scc1 %s" % skip)
    if host_is_check:
        a.e("s_lshr_b32 %s, %s, 29" % (T3, OP[1]))
        a.e("s_cmp_eq_u32 %s, %d" % (T3, K_ACC))
        a.e("s_cbranch_scc0 %s" % skip)
    a.e("s_and_b32 %s, %s, 0xff" % (T3, NXW[0]))
    a.e("s_cmp_lt_u32 %s, 16" % T3)
    a.e("s_cbranch_scc1 %s" % skip)
    a.e("s_cmp_eq_u32 %s, %s" % (NXW[3], IN_ONE))
    a.e("s_cbranch_scc1 %s" % skip)
    if not host_is_check:
        for w in (NXW[3], NXW[2]):
            a.e("s_cmp_eq_u32 %s, %s" % (w, OP[1]))
            a.e("s_cbranch_scc1 %s" % skip)
    fetch(a, NXW[3], VBB, AD_B, acc_label=after_b, stem=stem + "b", tag=TAG_B, vmf=2)
    a.label(after_b)
    fetch(a, NXW[2], VA, AD_A, acc_label=skip, stem=stem + "a", tag=TAG_A, vmf=2)
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
    descriptor(1, IN_CST, lambda w: a.e("s_lshl_b32 %s, %s, 5" % (w, IN_NCST)))
    descriptor(2, IN_GS, lambda w: a.e("s_mov_b32 %s, %s" % (w, IN_GSBYTES)))
    for j in range(8):
        a.e("v_mov_b32 %s, 0x%08x" % (P8[j], P_LIMBS[j]))
    for j in range(8):
        a.e("v_mov_b32 %s, 0" % ACC8[j])
    for j in range(9):
        a.e("s_mov_b32 %s, 0x%x" % (SP29[j], fr29.P_LIMBS[j]))
    a.e("s_mov_b32 %s, 0x%x" % (SNP29, fr29.NP))
    a.e("v_mov_b32 %s, -1" % FIRST)
    a.e("s_mov_b32 %s, -1" % TAG_A)
    a.e("s_mov_b32 %s, -1" % TAG_B)
    a.e("s_mov_b32 %s, -1" % TAG_V)
    a.e("s_mov_b32 %s, 0" % VMF)
    a.e("s_mov_b64 %s, 0" % UNSAT)
    a.e("s_mov_b32 %s, 0" % KCHK)
    a.e("s_mov_b32 %s, 0x%08x" % (SNP0, NP0))
    a.e("s_mov_b32 %s, %s" % (REM, IN_NOPS))
    a.e("s_cmp_eq_u32 %s, 0" % REM)
    a.e("s_cbranch_scc1 .Lzk_end")
    a.e("s_buffer_load_dwordx4 %s, %s, 0" % (NX, D_OPS))
    a.e("s_mov_b32 %s, 16" % PCOFF)

    # ---- per-op loop --------------------------------------------------------
    a.label(".Lzk_top")
    a.e("s_waitcnt lgkmcnt(0)")
    a.e("s_mov_b64 %s, %s" % (OP_LO, NX_LO))
    a.e("s_mov_b64 %s, %s" % (OP_HI, NX_HI))
    a.e("s_sub_u32 %s, %s, 1" % (REM, REM))
    a.e("s_cmp_eq_u32 %s, 0" % REM)
    a.e("s_cbranch_scc1 .Lzk_nopf")
    a.e("s_buffer_load_dwordx4 %s, %s, %s" % (NX, D_OPS, PCOFF))
    a.e("s_add_u32 %s, %s, 16" % (PCOFF, PCOFF))
    a.label(".Lzk_nopf")
    a.e("s_and_b32 %s, %s, 0xff" % (CODE, OP[0]))
    a.e("s_lshr_b32 %s, %s, 1" % (VMF, VMF))  # loads sent for during the last op are this op's loads now
    a.e("s_cmp_lt_u32 %s, 16" % CODE)
    a.e("s_cbranch_scc1 .Lzk_unary")

    # A CHECK whose right factor is the constant 1 (the constraint of a signal with a linear value, `value * 1 ==
    # signal`) multiplies nothing: the left factor is compared as it is.  x * 1 is x bit for bit, so no result changes.
    a.e("s_cmp_eq_u32 %s, %s" % (OP[3], IN_ONE))
    a.e("s_cbranch_scc1 .Lzk_b_is_one")  # rare: the common path falls through
    a.label(".Lzk_b_not_one")
    # binary ops and CHECK: fetch b, then a (or note that a is the accumulator)
    fetch(a, OP[3], VBB, AD_B, stem="b", tag=TAG_B)
    a.e("s_mov_b32 %s, 0" % AACC)
    fetch(a, OP[2], VA, AD_A, acc_label=".Lzk_a_is_acc", stem="a", tag=TAG_A)
    a.e("s_branch .Lzk_have_ab")
    a.label(".Lzk_a_is_acc")
    a.e("s_mov_b32 %s, 1" % AACC)
    a.label(".Lzk_have_ab")
    wait_loads(a)
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_MUL))
    a.e("s_cbranch_scc1 .Lzk_mul")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_ADD))
    a.e("s_cbranch_scc1 .Lzk_add")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_SUB))
    a.e("s_cbranch_scc1 .Lzk_sub")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_CHECK))
    a.e("s_cbranch_scc1 .Lzk_check")
    a.e("s_branch .Lzk_next")  # an opcode this stream does not know: no effect

    a.label(".Lzk_mul")
    a.e("s_cmp_eq_u32 %s, 1" % AACC)
    a.e("s_cbranch_scc1 .Lzk_mul_acc")
    a.label(".Lzk_mul_ab")
    mul_stream(a, A8, B8, ACC8, after_unpack=lambda: prefetch_next(a, "pm", False))
    a.e("s_branch .Lzk_store")
    a.label(".Lzk_mul_acc")
    mul_stream(a, ACC8, B8, ACC8, after_unpack=lambda: prefetch_next(a, "pn", False))
    a.e("s_branch .Lzk_store")

    a.label(".Lzk_add")
    a.e("s_cmp_eq_u32 %s, 1" % AACC)
    a.e("s_cbranch_scc1 .Lzk_add_acc")
    add_stream(a, A8, B8, ACC8)
    a.e("s_branch .Lzk_store")
    a.label(".Lzk_add_acc")
    add_stream(a, ACC8, B8, ACC8)
    a.e("s_branch .Lzk_store")

    a.label(".Lzk_sub")
    a.e("s_cmp_eq_u32 %s, 1" % AACC)
    a.e("s_cbranch_scc1 .Lzk_sub_acc")
    a.label(".Lzk_sub_ab")
    sub_stream(a, A8, B8, ACC8)
    a.e("s_branch .Lzk_store")
    a.label(".Lzk_sub_acc")
    sub_stream(a, ACC8, B8, ACC8)
    a.e("s_branch .Lzk_store")

    # CHECK: product into scratch registers (the a and b registers keep what their tags say), then compare with c
    a.label(".Lzk_check")
    a.e("s_cmp_eq_u32 %s, 1" % AACC)
    a.e("s_cbranch_scc1 .Lzk_check_acc")
    mul_stream(a, A8, B8, W8, after_unpack=lambda: prefetch_next(a, "pc", True))
    a.e("s_branch .Lzk_check_cmp")
    a.label(".Lzk_check_acc")
    mul_stream(a, ACC8, B8, W8, after_unpack=lambda: prefetch_next(a, "pd", True))
    a.e("s_branch .Lzk_check_cmp")
    a.label(".Lzk_b_is_one")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_CHECK))
    a.e("s_cbranch_scc0 .Lzk_b_not_one")
    a.label(".Lzk_check_one")  # the "product" is the left factor
    fetch(a, OP[2], VA, AD_A, acc_label=".Lzk_one_acc", stem="p", tag=TAG_A)
    wait_loads(a)
    for j in range(8):
        a.e("v_mov_b32 %s, %s" % (W8[j], A8[j]))
    a.e("s_branch .Lzk_check_cmp")
    a.label(".Lzk_one_acc")
    for j in range(8):
        a.e("v_mov_b32 %s, %s" % (W8[j], ACC8[j]))
    a.label(".Lzk_check_cmp")
    a.e("s_and_b32 %s, %s, 2" % (VMF, VMF))  # this op's loads have been waited for; loads for the next op may be under way
    fetch(a, OP[1], VBB, AD_C, acc_label=".Lzk_cmp_acc", stem="c", tag=TAG_B)
    wait_loads(a)

    def compare(c):
        a.e("v_xor_b32 %s, %s, %s" % (VT0, W8[0], c[0]))
        a.e("v_xor_b32 %s, %s, %s" % (VT1, W8[1], c[1]))
        for j in range(2, 8, 2):
            a.e("v_xor_b32 %s, %s, %s" % (E(0), W8[j], c[j]))
            a.e("v_xor_b32 %s, %s, %s" % (M, W8[j + 1], c[j + 1]))
            a.e("v_or_b32 %s, %s, %s" % (VT0, VT0, E(0)))
            a.e("v_or_b32 %s, %s, %s" % (VT1, VT1, M))
        a.e("v_or_b32 %s, %s, %s" % (VT0, VT0, VT1))
        a.e("s_nop 0")
        a.e("v_cmp_ne_u32 vcc, 0, %s" % VT0)
    compare(B8)
    a.e("s_branch .Lzk_cmp_count")
    a.label(".Lzk_cmp_acc")
    compare(ACC8)
    a.label(".Lzk_cmp_count")
    a.e("s_nop 1")
    a.e("s_and_b64 %s, vcc, %s" % (CC2, IN_LIVE))
    a.e("s_cbranch_scc0 .Lzk_cmp_ok")
    a.e("s_bcnt1_i32_b64 %s, %s" % (T0, CC2))
    a.e("s_add_u32 %s, %s, %s" % (UNSAT_LO, UNSAT_LO, T0))
    a.e("s_addc_u32 %s, %s, 0" % (UNSAT_HI, UNSAT_HI))
    # the constraint number: carried by the op (number + 1 in bits 8..31) or, when that is 0, the count so far
    a.e("s_lshr_b32 %s, %s, 8" % (T1, OP[0]))
    a.e("s_cbranch_scc1 .Lzk_cmp_id")
    a.e("s_add_u32 %s, %s, 1" % (T1, KCHK))
    a.label(".Lzk_cmp_id")
    a.e("s_sub_u32 %s, %s, 1" % (T1, T1))
    a.e("v_mov_b32 %s, %s" % (VT0, T1))
    a.e("v_mov_b32 %s, -1" % VT1)
    a.e("v_cndmask_b32_e64 %s, %s, %s, %s" % (VT0, VT1, VT0, CC2))
    a.e("v_min_u32 %s, %s, %s" % (FIRST, FIRST, VT0))
    a.label(".Lzk_cmp_ok")
    a.e("s_add_u32 %s, %s, 1" % (KCHK, KCHK))
    a.e("s_branch .Lzk_next")

    # ---- INPUT, OUTPUT, COPY, NEG ------------------------------------------
    a.label(".Lzk_unary")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_OUTPUT_SCALED))
    a.e("s_cbranch_scc1 .Lzk_output")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_OUTPUT))
    a.e("s_cbranch_scc1 .Lzk_output")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_INPUT))
    a.e("s_cbranch_scc1 .Lzk_input")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_INPUT_SCALED))
    a.e("s_cbranch_scc1 .Lzk_input")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_COPY))
    a.e("s_cbranch_scc1 .Lzk_copy")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_NEG))
    a.e("s_cbranch_scc1 .Lzk_neg")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_BARRIER))
    a.e("s_cbranch_scc0 .Lzk_next")  # NOP and anything unknown
    # BARRIER: everything this wavefront stored is visible to the workgroup once its memory
    # operations have completed; then wait for the other strands
    a.e("s_cmp_eq_u32 %s, -1" % IN_ROT)
    a.e("s_cbranch_scc0 .Lzk_barrier_turn")
    a.e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    a.e("s_barrier")
    a.e("s_branch .Lzk_next")
    # Priority turns, as in the stream on 29-bit limbs: the workgroups resident on a CU hold four different issue
    # priorities that turn with the constant-rate clock, each workgroup in its own phase, so that they finish
    # together instead of in the order they were placed (a SIMD issues from its oldest wavefront first).
    a.label(".Lzk_barrier_turn")
    a.e("s_memrealtime %s" % NOW)
    a.e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    a.e("s_barrier")
    a.e("s_lshr_b32 %s, %s, %s" % (T0, NOW_LO, IN_ROTSH))
    a.e("s_add_u32 %s, %s, %s" % (T0, T0, IN_ROT))
    a.e("s_and_b32 %s, %s, 3" % (T0, T0))
    a.e("s_cmp_lt_u32 %s, 2" % T0)
    a.e("s_cbranch_scc1 .Lzk_prio01")
    a.e("s_cmp_eq_u32 %s, 2" % T0)
    a.e("s_cbranch_scc1 .Lzk_prio2")
    a.e("s_setprio 3")
    a.e("s_branch .Lzk_next")
    a.label(".Lzk_prio2")
    a.e("s_setprio 2")
    a.e("s_branch .Lzk_next")
    a.label(".Lzk_prio01")
    a.e("s_cmp_eq_u32 %s, 0" % T0)
    a.e("s_cbranch_scc1 .Lzk_prio0")
    a.e("s_setprio 1")
    a.e("s_branch .Lzk_next")
    a.label(".Lzk_prio0")
    a.e("s_setprio 0")
    a.e("s_branch .Lzk_next")

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

    a.label(".Lzk_input")
    a.e("s_and_b32 %s, %s, 0x1fffffff" % (T1, OP[2]))
    io_descriptor(IN_INB, T1)
    limb_offsets()
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_INPUT_SCALED))
    a.e("s_cbranch_scc1 .Lzk_input_scaled")
    for j in range(8):
        a.e("buffer_load_dword %s, %s, %s, %s offen" % (A8[j], IN_VVEC, D_IO, cst_word(j)))
    for j in range(8):
        a.e("v_mov_b32 %s, 0x%08x" % (B8[j], R2_LIMBS[j]))
    a.e("s_mov_b32 %s, -1" % TAG_A)
    a.e("s_mov_b32 %s, -1" % TAG_B)
    a.e("s_waitcnt vmcnt(0)")
    a.e("s_branch .Lzk_mul_ab")
    # the input times 2^-256 is held as the input's own words: loaded into the accumulator, no product
    a.label(".Lzk_input_scaled")
    for j in range(8):
        a.e("buffer_load_dword %s, %s, %s, %s offen" % (ACC8[j], IN_VVEC, D_IO, cst_word(j)))
    a.e("s_waitcnt vmcnt(0)")
    a.e("s_branch .Lzk_store")

    a.label(".Lzk_output")
    a.e("s_cmp_eq_u64 %s, 0" % IN_OUTB)  # nobody wants the outputs: skip the conversion too
    a.e("s_cbranch_scc1 .Lzk_next")
    fetch(a, OP[2], VA, AD_A, acc_label=".Lzk_out_acc", stem="o", tag=TAG_A)
    wait_loads(a)
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_OUTPUT_SCALED))
    a.e("s_cbranch_scc1 .Lzk_out_scaled")
    redc_stream(a, A8)
    a.e("s_branch .Lzk_out_store")
    a.label(".Lzk_out_scaled")  # the output times 2^256: the held words as they are
    for j in range(8):
        a.e("v_mov_b32 %s, %s" % (E(j), A8[j]))
    a.e("s_branch .Lzk_out_store")
    a.label(".Lzk_out_acc")
    a.e("s_cmp_eq_u32 %s, %d" % (CODE, OP_OUTPUT_SCALED))
    a.e("s_cbranch_scc1 .Lzk_out_acc_scaled")
    redc_stream(a, ACC8)
    a.e("s_branch .Lzk_out_store")
    a.label(".Lzk_out_acc_scaled")
    for j in range(8):
        a.e("v_mov_b32 %s, %s" % (E(j), ACC8[j]))
    a.label(".Lzk_out_store")
    io_descriptor(IN_OUTB, OP[1])
    limb_offsets()
    a.e("s_mov_b64 exec, %s" % IN_LIVE)
    # non-temporal: the launch never reads an output back, so its lines need not stay in the L2 beside the slot lines
    # (the slot stores stay plain: a circuit that runs over few vectors finds its slot lines still in the L2)
    for j in range(8):
        a.e("buffer_store_dword %s, %s, %s, %s offen nt" % (E(j), IN_VVEC, D_IO, cst_word(j)))
    a.e("s_mov_b64 exec, -1")
    a.e("s_branch .Lzk_next")

    a.label(".Lzk_copy")
    fetch(a, OP[2], VACC, AD_A, stem="k")
    wait_loads(a)
    a.e("s_branch .Lzk_store")

    a.label(".Lzk_neg")  # 0 - a
    fetch(a, OP[2], VBB, AD_B, stem="n", tag=TAG_B)
    a.e("s_mov_b32 %s, -1" % TAG_A)
    for j in range(8):
        a.e("v_mov_b32 %s, 0" % A8[j])
    wait_loads(a)
    a.e("s_branch .Lzk_sub_ab")

    # ---- store the accumulator where the op says ---------------------------
    a.label(".Lzk_store")
    # a value remembered in the operand registers dies when its slot is overwritten
    a.e("s_cmp_eq_u32 %s, %s" % (OP[1], TAG_A))
    a.e("s_cselect_b32 %s, -1, %s" % (TAG_A, TAG_A))
    a.e("s_cmp_eq_u32 %s, %s" % (OP[1], TAG_B))
    a.e("s_cselect_b32 %s, -1, %s" % (TAG_B, TAG_B))
    a.e("s_cmp_eq_u32 %s, %s" % (OP[1], TAG_V))
    a.e("s_cselect_b32 %s, -1, %s" % (TAG_V, TAG_V))
    a.e("s_lshr_b32 %s, %s, 29" % (T0, OP[1]))
    a.e("s_and_b32 %s, %s, 0x1fffffff" % (T1, OP[1]))
    a.e("s_lshl_b32 %s, %s, 13" % (T1, T1))
    a.e("s_cmp_eq_u32 %s, %d" % (T0, K_LSLOT))
    a.e("s_cbranch_scc1 .Lzk_st_lds")
    a.e("s_cmp_eq_u32 %s, %d" % (T0, K_XSLOT))
    a.e("s_cbranch_scc1 .Lzk_st_x")
    a.e("s_cmp_eq_u32 %s, %d" % (T0, K_GSLOT))
    a.e("s_cbranch_scc0 .Lzk_next")
    a.e("v_add_u32 %s, %s, %s" % (AD_C, T1, IN_VLO))
    a.e("buffer_store_dwordx4 %s, %s, %s, 0 offen" % (quad(VACC, 0), AD_C, D_GS))
    a.e("buffer_store_dwordx4 %s, %s, %s, 0 offen offset:1024" % (quad(VACC, 1), AD_C, D_GS))
    a.e("s_nop 1")
    a.e("s_branch .Lzk_next")
    a.label(".Lzk_st_lds")
    a.e("s_add_u32 %s, %s, %s" % (T1, T1, IN_LDSB))
    a.e("v_add_u32 %s, %s, %s" % (AD_C, T1, IN_VLO))
    a.e("ds_write_b128 %s, %s" % (AD_C, quad(VACC, 0)))
    a.e("ds_write_b128 %s, %s offset:1024" % (AD_C, quad(VACC, 1)))
    a.e("s_branch .Lzk_next")
    a.label(".Lzk_st_x")
    a.e("s_lshr_b32 %s, %s, 2" % (T1, T1))  # index << 13 -> index << 11
    a.e("s_add_u32 %s, %s, %s" % (T1, T1, IN_XBASE))
    a.e("v_add_u32 %s, %s, %s" % (AD_C, T1, IN_VX))
    a.e("ds_write_b128 %s, %s" % (AD_C, quad(VACC, 0)))
    a.e("ds_write_b128 %s, %s offset:1024" % (AD_C, quad(VACC, 1)))
    a.label(".Lzk_next")
    a.e("s_cmp_lg_u32 %s, 0" % REM)
    a.e("s_cbranch_scc1 .Lzk_top")

    a.label(".Lzk_end")
    a.e("s_setprio 0")
    a.e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    a.e("s_mov_b64 %s, %s" % (OUT_UNSAT, UNSAT))
    a.e("v_mov_b32 %s, %s" % (OUT_FIRST, FIRST))
    return a.lines


def clobbers():
    v = ['"v%d"' % k for k in range(VB, V_LAST + 1)]
    s = ['"s%d"' % k for k in range(S_FIRST, S_LAST + 1)]
    return ", ".join(v + s + ['"vcc"', '"scc"', '"memory"'])


def render(lines):
    out = ["// generated by tools/gen_interp_asm.py -- do not edit",
           "// vector registers v%d..v%d, scalar registers s%d..s%d" % (VB, V_LAST, S_FIRST, S_LAST),
           "#define ZKIR_INTERP_CORE_ASM \\"]
    for l in lines:
        out.append('  "%s\\n" \\' % l)
    out.append("")
    out.append("#define ZKIR_INTERP_CORE_CLOBBERS %s" % clobbers())
    out.append("#define ZKIR_INTERP_CORE_VGPR_LAST %d" % V_LAST)
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "zkir_amd", "csrc", "hip", "interp_core_asm.inc")
    lines = build()
    with open(path, "w") as f:
        f.write(render(lines))
    n_valu = sum(1 for l in lines if l.startswith("v_"))
    print("wrote %s: %d lines, %d vector ALU instructions" % (path, len(lines), n_valu))
