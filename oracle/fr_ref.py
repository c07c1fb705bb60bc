"""BN254 scalar-field operator semantics with Python big integers.

TEST INFRASTRUCTURE ONLY: nothing under oracle/ is part of the shipped path; only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

PARITY UNPINNED: the reference (razn-v/zkir) has no field arithmetic at all
(`Field` is a `String`, src/ast.rs:12) and holds no test vectors; the arithmetic
lives in iden3/circom, which the reference only targets through the operator
tokens it prints (src/circom.rs:35-157) and which is neither vendored nor
version-pinned (Cargo.lock:5-7 lists no dependency).  The definitions below
restate circom's published operator semantics ("Basic operators" in the circom
documentation; circom_algebra's modular_arithmetic) from memory; they could not
be run against circom here.  What these functions DO pin is the GPU kernels and
the C oracle against an independent big-integer implementation.
"""

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
HALF = P // 2
MASK254 = (1 << 254) - 1
MASK256 = (1 << 256) - 1
R = (1 << 256) % P
R2 = (R * R) % P
N0_32 = (-pow(P, -1, 1 << 32)) % (1 << 32)
N0_64 = (-pow(P, -1, 1 << 64)) % (1 << 64)


def limbs32(x):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


def from_limbs32(l):
    return sum(int(v) << (32 * i) for i, v in enumerate(l))


def signed(x):
    """circom compares the representative in (-p/2, p/2]."""
    return x - P if x > HALF else x


class Fault(Exception):
    """Division by zero; the evaluators return 0 and count a fault."""


# one function per Expr variant; the token is the one src/circom.rs prints
def add(a, b):  # `+`  circom.rs:35-41
    return (a + b) % P


def sub(a, b):  # `-`  circom.rs:42-46
    return (a - b) % P


def mul(a, b):  # `*`  circom.rs:47-51
    return (a * b) % P


def power(a, b):  # `**` circom.rs:52-56 (exponent = unsigned representative)
    return pow(a, b, P)


def div(a, b):  # `/`  circom.rs:57-61
    if b == 0:
        raise Fault()
    return (a * pow(b, -1, P)) % P


def idiv(a, b):  # `\`  circom.rs:62-66
    if b == 0:
        raise Fault()
    return a // b


def rem(a, b):  # `%`  circom.rs:67-71
    if b == 0:
        raise Fault()
    return a % b


def lt(a, b):  # circom.rs:72-78
    return int(signed(a) < signed(b))


def le(a, b):  # circom.rs:79-85
    return int(signed(a) <= signed(b))


def gt(a, b):  # circom.rs:86-92
    return int(signed(a) > signed(b))


def ge(a, b):  # circom.rs:93-99
    return int(signed(a) >= signed(b))


def eq(a, b):  # circom.rs:100-106
    return int(a == b)


def land(a, b):  # `&&` circom.rs:107-113
    return int(a != 0 and b != 0)


def lor(a, b):  # `||` circom.rs:114-120
    return int(a != 0 or b != 0)


def band(a, b):  # circom.rs:122-128
    return (a & b) % P


def bor(a, b):  # circom.rs:129-135
    return (a | b) % P


def bxor(a, b):  # circom.rs:137-143
    return (a ^ b) % P


def bnot(a):  # `~` circom.rs:136: complement over the 254 bits of p, then mod p
    return ((~a) & MASK254) % P


def bnot256(a):  # alternative reading of `~`: complement over 256 bits, then mod p
    return ((~a) & MASK256) % P


def shr(a, k):  # circom.rs:144-150
    if k <= HALF:
        return 0 if k >= 254 else a >> k
    return shl(a, P - k)


def shl(a, k):  # circom.rs:151-157
    if k <= HALF:
        return 0 if k >= 254 else ((a << k) & MASK254) % P
    return shr(a, P - k)


def neg(a):
    return (-a) % P


# opcode numbers of include/zkir_hip.h
OP_NOP, OP_INPUT, OP_OUTPUT, OP_COPY, OP_NEG, OP_BNOT, OP_BNOT256, OP_BARRIER = 0, 1, 2, 3, 4, 5, 6, 7
OP_INPUT_SCALED, OP_OUTPUT_SCALED = 8, 9  # the input times 2^-256, the output times 2^256 (include/zkir_hip.h)
SCALE = pow(2, 256, P)
SCALE_INV = pow(SCALE, -1, P)
(OP_ADD, OP_SUB, OP_MUL, OP_POW, OP_DIV, OP_IDIV, OP_REM, OP_LT, OP_LE, OP_GT, OP_GE, OP_EQ, OP_AND, OP_OR,
 OP_BAND, OP_BOR, OP_BXOR, OP_SHR, OP_SHL) = range(16, 35)
OP_CHECK = 48

BINARY = {
    OP_ADD: add, OP_SUB: sub, OP_MUL: mul, OP_POW: power, OP_DIV: div, OP_IDIV: idiv, OP_REM: rem,
    OP_LT: lt, OP_LE: le, OP_GT: gt, OP_GE: ge, OP_EQ: eq, OP_AND: land, OP_OR: lor,
    OP_BAND: band, OP_BOR: bor, OP_BXOR: bxor, OP_SHR: shr, OP_SHL: shl,
}
UNARY = {OP_COPY: lambda a: a, OP_NEG: neg, OP_BNOT: bnot, OP_BNOT256: bnot256}

OP_NAMES = {
    OP_NOP: "nop", OP_INPUT: "input", OP_OUTPUT: "output", OP_COPY: "copy", OP_NEG: "neg", OP_BNOT: "bnot",
    OP_BNOT256: "bnot256", OP_ADD: "add", OP_SUB: "sub", OP_MUL: "mul", OP_POW: "pow", OP_DIV: "div",
    OP_IDIV: "idiv", OP_REM: "rem", OP_LT: "lt", OP_LE: "le", OP_GT: "gt", OP_GE: "ge", OP_EQ: "eq",
    OP_AND: "and", OP_OR: "or", OP_BAND: "band", OP_BOR: "bor", OP_BXOR: "bxor", OP_SHR: "shr", OP_SHL: "shl",
    OP_CHECK: "check", OP_INPUT_SCALED: "input_scaled", OP_OUTPUT_SCALED: "output_scaled",
}

KIND_SHIFT = 29
INDEX_MASK = 0x1FFFFFFF
K_GSLOT, K_LSLOT, K_CONST, K_INPUT, K_ACC, K_NONE, K_XSLOT = 0, 1, 2, 3, 4, 5, 6


def operand(kind, idx=0):
    return (kind << KIND_SHIFT) | (idx & INDEX_MASK)


def eval_tape_one(ops, consts, inputs, strand_ops=None):
    """Interpret a tape for ONE input vector with big integers.

    ops: iterable of (code, dst, a, b); consts / inputs: lists of ints in [0, p).
    strand_ops: for a strand tape, the number of ops of each strand (ops holds the strands one
    after the other).  The strands have private accumulators and G / L slots, share the exchange
    slots, and are run here one after the other from barrier to barrier; constraint numbers
    then come from the CHECK ops ((code >> 8) - 1) instead of their position.
    Returns (outputs dict, unsat constraint indices in ascending order, faults).
    """
    ops = list(ops)
    counts = list(strand_ops) if strand_ops else [len(ops)]
    begin = [sum(counts[:i]) for i in range(len(counts) + 1)]
    state = [{"g": {}, "l": {}, "acc": 0, "pc": begin[i]} for i in range(len(counts))]
    xs = {}
    outputs, unsat, faults, k = {}, [], 0, 0

    def rd(st, o):
        kind, idx = o >> KIND_SHIFT, o & INDEX_MASK
        if kind == K_GSLOT:
            return st["g"][idx]
        if kind == K_LSLOT:
            return st["l"][idx]
        if kind == K_CONST:
            return consts[idx]
        if kind == K_ACC:
            return st["acc"]
        if kind == K_XSLOT and strand_ops:
            return xs[idx]
        raise ValueError("bad source operand kind %d" % kind)

    def wr(st, o, v):
        kind, idx = o >> KIND_SHIFT, o & INDEX_MASK
        if kind == K_GSLOT:
            st["g"][idx] = v
        elif kind == K_LSLOT:
            st["l"][idx] = v
        elif kind == K_XSLOT and strand_ops:
            xs[idx] = v
        elif kind != K_NONE:
            raise ValueError("bad destination operand kind %d" % kind)

    running = True
    while running:
        running = False
        for i, st in enumerate(state):
            while st["pc"] < begin[i + 1]:
                full, dst, a, b = ops[st["pc"]]
                st["pc"] += 1
                code = full & 0xFF
                if code == OP_NOP:
                    continue
                if code == OP_BARRIER:
                    if not strand_ops:
                        raise ValueError("BARRIER in a plain tape")
                    break
                if code in (OP_INPUT, OP_INPUT_SCALED):
                    assert a >> KIND_SHIFT == K_INPUT
                    st["acc"] = inputs[a & INDEX_MASK] if code == OP_INPUT else mul(inputs[a & INDEX_MASK], SCALE_INV)
                    wr(st, dst, st["acc"])
                elif code == OP_OUTPUT:
                    outputs[dst] = rd(st, a)
                elif code == OP_OUTPUT_SCALED:
                    outputs[dst] = mul(rd(st, a), SCALE)
                elif code == OP_CHECK:
                    if mul(rd(st, a), rd(st, b)) != rd(st, dst):
                        unsat.append((full >> 8) - 1 if full >> 8 else k)
                    k += 1
                elif code in UNARY:
                    st["acc"] = UNARY[code](rd(st, a))
                    wr(st, dst, st["acc"])
                elif code in BINARY:
                    try:
                        st["acc"] = BINARY[code](rd(st, a), rd(st, b))
                    except Fault:
                        st["acc"], faults = 0, faults + 1
                    wr(st, dst, st["acc"])
                else:
                    raise ValueError("bad opcode %d" % code)
            if st["pc"] < begin[i + 1]:
                running = True
    return outputs, sorted(unsat), faults
