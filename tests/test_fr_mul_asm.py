"""The generated gfx950 instruction stream of the Montgomery multiplier, executed
by a small Python model of the handful of instructions it uses, against big
integers.  Checks the algorithm and the register plan without a GPU (issue
hazards are covered by the GPU parity tests)."""
import os
import random
import re

from oracle import fr_ref as F

INC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zkir_amd", "csrc", "hip",
                   "fr_mul_asm.inc")
M32 = 0xFFFFFFFF


def load_stream(macro="ZKIR_FR_MUL_ASM"):
    lines, on = [], False
    for raw in open(INC):
        if raw.startswith("#define"):
            on = raw.split()[1] == macro
            continue
        m = re.match(r'\s*"(.*)\\n" \\', raw)
        if m and on:
            lines.append(m.group(1))
    assert lines, "no instructions parsed"
    return lines


def simulate(lines, a, b=None):
    reg = {}
    al, pl = F.limbs32(a), F.limbs32(F.P)
    if b is not None:  # multiplier operands: a %9.., b %17.., p %25.., np0 %33
        bl = F.limbs32(b)
        for j in range(8):
            reg["%%%d" % (9 + j)] = al[j]
            reg["%%%d" % (17 + j)] = bl[j]
            reg["%%%d" % (25 + j)] = pl[j]
        reg["%33"] = F.N0_32
    else:  # reduction operands: a %9.., p %17.., np0 %25
        for j in range(8):
            reg["%%%d" % (9 + j)] = al[j]
            reg["%%%d" % (17 + j)] = pl[j]
        reg["%25"] = F.N0_32
    vcc = 0

    wrote_output = [False]

    def rd(x):
        # results may share registers with inputs: no input may be read once a result is written
        assert not (wrote_output[0] and x.startswith("%")), "input %s read after a result was written" % x
        if x in reg:
            return reg[x]
        if re.fullmatch(r"\d+", x):
            return int(x)
        raise KeyError("read of unwritten register " + x)

    def rd64(x):
        if x == "0":
            return 0
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", x)
        return rd("v" + m.group(1)) | (rd("v" + m.group(2)) << 32)

    def wr64(x, v):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", x)
        reg["v" + m.group(1)] = v & M32
        reg["v" + m.group(2)] = (v >> 32) & M32

    last_vcc_writer = -10
    n = 0  # issue slot: every instruction takes one, `s_nop k` takes k + 1
    for ins in lines:
        op, _, rest = ins.partition(" ")
        args = [s.strip() for s in rest.split(",")] if rest else []
        n += 1
        if op == "s_nop":
            n += int(args[0])
            continue
        if args and re.fullmatch(r"%[0-7]", args[0]):
            v_ = None  # evaluate sources first, then mark
        if op == "v_mov_b32":
            reg[args[0]] = rd(args[1])
        elif op == "v_mad_u64_u32":
            v = rd(args[2]) * rd(args[3]) + rd64(args[4])
            assert v < 1 << 64, "multiply-add overflowed 64 bits"
            wr64(args[0], v)
        elif op == "v_mul_lo_u32":
            reg[args[0]] = (rd(args[1]) * rd(args[2])) & M32
        elif op in ("v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32"):
            assert args[1] == "vcc"
            cin = 0
            if op in ("v_addc_co_u32", "v_subb_co_u32"):
                assert args[4] == "vcc"
                # a carry may be consumed only after one other instruction has issued
                assert n - last_vcc_writer >= 2, "carry consumed back to back at line %d" % n
                cin = vcc
            if op.startswith("v_add"):
                v = rd(args[2]) + rd(args[3]) + cin
                vcc = v >> 32
            else:
                v = rd(args[2]) - rd(args[3]) - cin
                vcc = 1 if v < 0 else 0
            reg[args[0]] = v & M32
            last_vcc_writer = n
        elif op == "v_cndmask_b32":
            assert args[3] == "vcc" and n - last_vcc_writer >= 3
            reg[args[0]] = rd(args[2]) if vcc else rd(args[1])
        else:
            raise AssertionError("instruction not modelled: " + ins)
        if args and re.fullmatch(r"%[0-7]", args[0]):
            wrote_output[0] = True
    return sum(reg["%%%d" % j] << (32 * j) for j in range(8))


def test_stream_matches_big_integers():
    lines = load_stream()
    rnd = random.Random(5)
    rinv = pow(1 << 256, -1, F.P)
    edge = [0, 1, 2, F.P - 1, F.P - 2, F.HALF, F.HALF + 1, (1 << 64) - 1, 1 << 253, F.R, F.R2, (1 << 254) % F.P]
    cases = [(x, y) for x in edge for y in edge]
    cases += [(rnd.randrange(F.P), rnd.randrange(F.P)) for _ in range(300)]
    for a, b in cases:
        assert simulate(lines, a, b) == (a * b * rinv) % F.P


def test_stream_cost():
    lines = load_stream()
    valu = [l for l in lines if l.startswith("v_")]
    muls = [l for l in valu if l.startswith("v_mad_u64_u32") or l.startswith("v_mul_lo_u32")]
    assert len(muls) == 136  # 2*8*8 multiply-adds + 8 low products
    assert len(valu) <= 300


def test_reduction_stream_matches_big_integers():
    lines = load_stream("ZKIR_FR_REDC_ASM")
    rnd = random.Random(6)
    rinv = pow(1 << 256, -1, F.P)
    for a in [0, 1, 2, F.P - 1, F.HALF, F.R, F.R2, (1 << 253)] + [rnd.randrange(F.P) for _ in range(300)]:
        assert simulate(lines, a) == (a * rinv) % F.P
    assert len([l for l in lines if l.startswith("v_")]) <= 160
