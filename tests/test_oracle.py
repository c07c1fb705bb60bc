"""The checkers themselves, pinned before they are trusted: big-integer operator
semantics and the C tape interpreter against the committed known answers, and the
PRNG restatement against values derived by hand from src/rng.rs:18-24."""
import json
import os

import pytest

from oracle import c_oracle as C
from oracle import fr_ref as F
from oracle import zkir_ref as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def test_constants():
    # the field constants every implementation hard-codes
    assert F.P == 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    assert F.P.bit_length() == 254
    assert F.N0_32 == 0xEFFFFFFF and F.N0_64 == 0xC2E1F593EFFFFFFF
    assert F.limbs32(F.R) == [0x4ffffffb, 0xac96341c, 0x9f60cd29, 0x36fc7695, 0x7879462e, 0x666ea36f, 0x9a07df2f, 0x0e0a77c1]
    assert F.limbs32(F.R2) == [0xae216da7, 0x1bb8e645, 0xe35c59e3, 0x53fe3ab1, 0x53bb8085, 0x8c49833d, 0x7f4e44a5, 0x0216d0b1]
    assert F.limbs32(F.HALF) == [0xf8000000, 0xa1f0fac9, 0x3cdcb848, 0x9419f424, 0x40c0ac2e, 0xdc2822db, 0x7098d014, 0x18322739]
    assert (F.P - 1) % (1 << 28) == 0 and (F.P - 1) % (1 << 29) != 0  # two-adicity 28


def test_operator_identities():
    """Properties the definitions must satisfy whatever the fixtures say."""
    import random
    rnd = random.Random(3)
    for _ in range(200):
        a, b = rnd.randrange(F.P), rnd.randrange(1, F.P)
        assert F.mul(F.div(a, b), b) == a
        assert F.idiv(a, b) * b + F.rem(a, b) == a
        assert F.add(F.sub(a, b), b) == a
        assert F.lt(a, b) + F.ge(a, b) == 1 and F.gt(a, b) + F.le(a, b) == 1
        assert F.bxor(F.bxor(a, b) if (a ^ b) < F.P else 0, 0) >= 0
        k = rnd.randrange(0, 254)
        assert F.shr(a, k) == a >> k and F.shl(a, F.P - k) == F.shr(a, k)
        assert F.power(a, F.P - 1) == (1 if a else 0) or a == 0
    assert F.power(0, 0) == 1
    assert F.shr(5, 254) == 0 and F.shl(5, 300) == 0 and F.shl(1, 253) == (1 << 253) % F.P
    assert F.lt(F.P - 1, 0) == 1  # p-1 stands for -1
    assert F.bnot(0) == ((1 << 254) - 1) % F.P and F.bnot256(0) == ((1 << 256) - 1) % F.P


def test_fr_ref_against_fixture():
    kat = load("fr_kat.json")
    assert len(kat) == 23
    for name, ent in kat.items():
        code = ent["code"]
        for a, b, r, fault in ent["rows"]:
            a, b, r = int(a, 16), int(b, 16), int(r, 16)
            try:
                got, gf = (F.BINARY[code](a, b) if code in F.BINARY else F.UNARY[code](a)), 0
            except F.Fault:
                got, gf = 0, 1
            assert (got, gf) == (r, fault), (name, hex(a), hex(b))


def test_c_oracle_against_fixture():
    kat = load("fr_kat.json")
    for name, ent in kat.items():
        for a, b, r, fault in ent["rows"]:
            got, gf = C.op(ent["code"], int(a, 16), int(b, 16))
            assert (got, gf) == (int(r, 16), fault), (name, a, b)


def test_rng_hand_values():
    # first values written out from the definition in src/rng.rs:18-24 (next() returns the
    # state before the update; shifts 13 / 17 / 43 on 64 bits)
    hand = {
        0x1: [0x1, 0x100080000002001, 0x10000000000201, 0x101080800402221, 0x1000000040023],
        0x2A: [0x2a, 0x2a01400000054028, 0x2a0010008001428, 0x2a0b51400a85116a, 0x400a800100a0053e],
        0x9E3779B97F4A7C15: [0x9e3779b97f4a7c15, 0x7723e6d01be0c471, 0xccd7d050f1b8e5c6, 0xa847e5090a33d346,
                             0xb1bdb63c127cab6b],
    }
    for seed, vals in hand.items():
        r = R.Rng(seed)
        assert [r.next() for _ in range(5)] == vals
    # second value of seed 1 by hand: 1 ^ 1<<13 = 0x2001; >>17 = 0; ^ (0x2001 << 43) = 0x100080000002001
    assert (0x2001 ^ ((0x2001 << 43) & ((1 << 64) - 1))) == 0x100080000002001


def test_rng_fixture_and_rand_rules():
    kat = load("rng_kat.json")
    for seed, ent in kat.items():
        r = R.Rng(int(seed, 16))
        assert [hex(r.next()) for _ in range(16)] == ent["next"]
        r = R.Rng(int(seed, 16))
        assert r.rand_string(8) == ent["rand_string8"]
    r = R.Rng(99)
    before = r.s
    assert r.rand(7, 7) == 7 and r.s == before  # no draw when min == max (rng.rs:29-31)
    assert r.rand_vec([]) is None and r.s == before  # rng.rs:48-50
    assert r.rand_vec([5]) == 5 and r.s == before
    with pytest.raises(AssertionError):
        r.rand(3, 2)


def test_generator_restatement_against_fixture():
    gold = load("circuits.json")
    for seed, ent in gold["reference"].items():
        c = R.Generator(R.Rng(int(seed, 16))).generate()
        assert R.print_text(c) == ent["text"]
        assert R.transpile_circuit(c) == ent["circom"]
        assert R.print_text(R.parse_text(ent["text"])) == ent["text"]
    for seed, ent in gold["extended"].items():
        c = R.Generator(R.Rng(int(seed, 16)), **ent["config"]).generate()
        assert R.print_text(c) == ent["text"]


def test_reference_shape_invariants():
    """What src/generator.rs guarantees about any circuit it emits."""
    for seed in range(1, 60):
        c = R.Generator(R.Rng(seed)).generate()
        assert len(c["instrs"]) <= 30  # :147
        assert all(not any(v["init"]) for v in c["vars"])  # Circuit.variables is the stale copy (:84, :255-260)
        names = [v["name"] for v in c["vars"]]
        assert all(len(n) == 8 and n.isalpha() for n in names)  # :264, rng.rs:37

        def depth(e):
            return 0 if e[0] in ("const", "var") else 1 + max(depth(x) for x in e[1:])

        def walk(body):
            for i in body:
                if i[0] == "for":
                    assert i[2][0] == "le" and i[3][0] == "assign" and i[4]  # :422-448
                    start, bound = int(i[1][2][1]), int(i[2][2][1])
                    assert 0 <= start <= 30 and start <= bound <= 60
                    walk(i[4])
                else:
                    assert i[0] in ("decl", "assign", "aassign", "constrain")  # if / while are not dispatched (:190,192)
                    e = i[-1]
                    if e is not None and not isinstance(e, int):
                        assert depth(e) <= 1  # :486-488

        walk(c["instrs"])


def test_interpreter_hand_cases():
    txt = '''(circuit (vars (var 0 "a" field input (init 0)) (var 1 "b" field input (init 0))
      (var 2 "o" field output (init 0)) (var 3 "t" field local (init 0)) (var 4 "arr" (array 3) local (init 0 0 0)))
     (instrs (decl 0) (decl 1) (decl 2) (decl 3 (add (var 0) (const "5"))) (decl 4)
       (aassign 4 (const "1") (mul (var 3) (const "2")))
       (constrain 2 (add (mul (var 3) (var 1)) (var (4 1))))))'''
    c = R.parse_text(txt)
    it = R.Interpreter(c, [7, 11]).run()
    assert it.outputs == [(7 + 5) * 11 + (7 + 5) * 2]
    assert it.constraints == [(12, 11, 12 * 11)] and it.unsat() == []
    # classes circom would refuse
    cases = {
        "NonQuadratic": '(constrain 2 (mul (mul (var 0) (var 1)) (var 0)))',
        "DoubleAssign": '(constrain 2 (var 0)) (constrain 2 (var 1))',
        "OOBIndex": '(constrain 2 (var (4 3)))',
        "DivByZero": '(constrain 2 (div (var 0) (const "0")))',
        "UnknownControl": '(if (var 0) (then (constrain 2 (var 1))))',
        "UninitSignal": '(constrain 2 (var 2))',
        "LoopCap": '(while (const "1") (body (assign 3 (const "1"))))',
    }
    for cls, stmt in cases.items():
        t = txt.replace('(constrain 2 (add (mul (var 3) (var 1)) (var (4 1))))', stmt)
        with pytest.raises(R.Rejected) as ei:
            R.Interpreter(R.parse_text(t), [7, 11], max_iterations=100).run()
        assert ei.value.cls == cls
