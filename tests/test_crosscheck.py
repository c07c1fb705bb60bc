"""The circom cross-check harness: file formats (round trips and fixed byte layouts), input.json
naming, output and constraint comparison.  No circom here: its side of the exchange is played by
files this module writes itself."""
import json
import os
import random
import struct

import pytest

import zkir_amd as z
from oracle import fr_ref as F
from zkir_amd import crosscheck as X


def test_wtns_layout_and_round_trip():
    vals = [1, 5, F.P - 1, 0]
    data = X.write_wtns(vals)
    assert data[:12] == b"wtns" + struct.pack("<II", 2, 2)
    assert struct.unpack_from("<IQ", data, 12) == (1, 4 + 32 + 4)          # header section
    assert struct.unpack_from("<I", data, 24) == (32,)
    assert int.from_bytes(data[28:60], "little") == F.P
    assert struct.unpack_from("<I", data, 60) == (4,)
    assert struct.unpack_from("<IQ", data, 64) == (2, 4 * 32)              # data section
    assert X.read_wtns(data) == (F.P, vals)
    with pytest.raises(ValueError):
        X.read_wtns(data[:-5])
    with pytest.raises(ValueError):
        X.read_wtns(b"r1cs" + data[4:])


def test_r1cs_round_trip_and_constraint_check():
    # wires: 0 = one, 1 = out, 2 = a, 3 = b, 4 = c;   (a + 1) * b == out - c   and   a * a == a*a (wire 5)
    cons = [({2: 1, 0: 1}, {3: 1}, {1: 1, 4: F.P - 1}), ({2: 1}, {2: 1}, {5: 1})]
    data = X.write_r1cs(6, 1, 0, 3, cons)
    r = X.read_r1cs(data)
    assert (r["prime"], r["n_wires"], r["n_pub_out"], r["n_pub_in"], r["n_prv_in"]) == (F.P, 6, 1, 0, 3)
    assert r["constraints"] == [tuple(dict(lc) for lc in c) for c in cons]
    assert r["wire2label"] == list(range(6))
    rnd = random.Random(3)
    a, b, c = (rnd.randrange(F.P) for _ in range(3))
    out = ((a + 1) * b + c) % F.P
    wit = [1, out, a, b, c, a * a % F.P]
    assert X.failed_constraints(r, wit) == []
    wit[1] = (out + 1) % F.P
    assert X.failed_constraints(r, wit) == [0]
    wit[5] ^= 2
    assert X.failed_constraints(r, wit) == [0, 1]


def test_sym_reader():
    sym = X.read_sym("1,1,0,main.out\n2,2,0,main.a\n3,-1,0,main.tmp\n4,3,0,main.arr[0]\n")
    assert sym == {"main.out": 1, "main.a": 2, "main.tmp": -1, "main.arr[0]": 3}


def _circuit(seed):
    g = z.default_generator_config()
    g.min_instructions = g.max_instructions = 40
    g.forced_inputs, g.inputs_readable, g.arithmetic_only, g.quadratic_aware = 3, 1, 1, 1
    g.no_loops, g.safe_array_index = 1, 1
    c = z.Generator(seed, g).generate()
    return c, z.CircomTranspiler(c).transpile_circuit(), z.Lowerer(c).lower()


def test_emit_and_compare(tmp_path):
    circuit, text, tape = _circuit(21)
    rnd = random.Random(4)
    vectors = [[rnd.randrange(F.P) for _ in range(tape.hdr.n_inputs)] for _ in range(3)]
    # this evaluator's side is played by the big-integer tape interpreter of the oracle
    outs = []
    for v in vectors:
        o, unsat, _ = F.eval_tape_one(tape.ops(), tape.consts(), v)
        assert not unsat
        outs.append([o.get(k, 0) for k in range(tape.hdr.n_outputs)])
    d = str(tmp_path / "case")
    X.emit_case(d, circuit, text, tape, vectors, outs)
    assert open(os.path.join(d, "circuit.circom")).read() == text
    doc = json.load(open(os.path.join(d, "input_1.json")))
    declared = [l.split()[2].rstrip(";") for l in text.splitlines() if l.strip().startswith("signal input")]
    assert sorted(doc) == sorted(n.split("[")[0] for n in declared)
    assert [int(doc[n]) for n in declared] == vectors[1]
    exp = json.load(open(os.path.join(d, "expected_2.json")))
    assert len(exp) == tape.hdr.n_outputs
    # circom's side: outputs on wires 1..n in declaration order, then the inputs
    names = list(exp)
    sym_text = "".join("%d,%d,0,main.%s\n" % (i + 1, i + 1, n) for i, n in enumerate(names))
    with open(os.path.join(d, "circuit.sym"), "w") as f:
        f.write(sym_text)
    for k in range(3):
        e = json.load(open(os.path.join(d, "expected_%d.json" % k)))
        wit = [1] + [int(e[n]) for n in names] + vectors[k]
        if k == 1:
            wit[2] = (wit[2] + 1) % F.P  # circom "disagrees" on the second output of vector 1
        with open(os.path.join(d, "witness_%d.wtns" % k), "wb") as f:
            f.write(X.write_wtns(wit))
    rep = X.compare_case(d)
    assert list(rep) == [1]
    assert rep[1]["outputs"] == [("main." + names[1], int(json.load(open(os.path.join(d, "expected_1.json")))[names[1]]),
                                  (int(json.load(open(os.path.join(d, "expected_1.json")))[names[1]]) + 1) % F.P)]
    assert X.main(["compare", d]) == 1


def test_emit_cli_without_gpu(tmp_path):
    # the reference's own generator settings: this seed is refused (by circom too, one expects)
    d = str(tmp_path / "ref")
    assert X.main(["emit", "--seed", "2", "--vectors", "2", "--out", d, "--no-evaluate"]) == 0
    verdict = json.load(open(os.path.join(d, "verdict.json")))
    assert not verdict["accepted"] and verdict["class"] == "DoubleAssign"
    assert sorted(os.listdir(d)) == ["circuit.circom", "verdict.json"]
    # the extended settings give circuits that lower
    d = str(tmp_path / "ext")
    assert X.main(["emit", "--seed", "2", "--vectors", "2", "--out", d, "--no-evaluate", "--extended"]) == 0
    assert json.load(open(os.path.join(d, "verdict.json")))["accepted"]
    assert sorted(os.listdir(d)) == ["circuit.circom", "input_0.json", "input_1.json", "run.sh", "verdict.json"]


def test_verdict_comparison(tmp_path):
    d = str(tmp_path / "ref")
    assert X.main(["emit", "--seed", "2", "--vectors", "1", "--out", d, "--no-evaluate"]) == 0  # DoubleAssign here
    with open(os.path.join(d, "circom.log"), "w") as f:
        f.write("error[T3001]: Exception caused by invalid assignment: signal already assigned\n")
    assert X.compare_verdict(d) == ("DoubleAssign", "DoubleAssign") and X.compare_case(d) == {}
    with open(os.path.join(d, "circom.log"), "w") as f:
        f.write("template instances: 1\nEverything went okay\n")
    assert X.compare_case(d) == {"verdict": ["DoubleAssign", "Ok"]}
    assert X.classify_circom_log("error[T3001]: Non quadratic constraints are not allowed!") == "NonQuadratic"
    assert X.classify_circom_log("error[P1000]: something new") == "Other"


# ---- byte strings assembled by hand from iden3's published layouts (not through this module's writers) ----
P_LE = bytes.fromhex("010000f093f5e1439170b97948e833285d588181b64550b829a031e1724e6430")  # p, 32 bytes little-endian


def _fe(v):
    return bytes([(v >> (8 * i)) & 0xFF for i in range(32)])


def test_hand_assembled_wtns_bytes():
    """wtns v2: "wtns", version, section count; section 1 = {n8, prime, witness count}, section 2 = the values."""
    assert int.from_bytes(P_LE, "little") == F.P
    vals = [1, 33, F.P - 2, 1 << 200]
    hdr = bytes([32, 0, 0, 0]) + P_LE + bytes([4, 0, 0, 0])
    body = b"".join(_fe(v) for v in vals)
    data = (b"wtns" + bytes([2, 0, 0, 0]) + bytes([2, 0, 0, 0])
            + bytes([1, 0, 0, 0]) + bytes([40, 0, 0, 0, 0, 0, 0, 0]) + hdr
            + bytes([2, 0, 0, 0]) + bytes([128, 0, 0, 0, 0, 0, 0, 0]) + body)
    assert len(data) == 12 + 12 + 40 + 12 + 128
    assert X.read_wtns(data) == (F.P, vals)
    assert X.write_wtns(vals) == data  # the module's writer produces the same bytes
    # sections in the other order (the container allows any)
    swapped = (b"wtns" + bytes([2, 0, 0, 0, 2, 0, 0, 0])
               + bytes([2, 0, 0, 0, 128, 0, 0, 0, 0, 0, 0, 0]) + body
               + bytes([1, 0, 0, 0, 40, 0, 0, 0, 0, 0, 0, 0]) + hdr)
    assert X.read_wtns(swapped) == (F.P, vals)


def test_hand_assembled_r1cs_bytes():
    """r1cs v1: header section {n8, prime, nWires, nPubOut, nPubIn, nPrvIn, nLabels (u64), mConstraints},
    constraint section {per linear combination: term count, (wire u32, coefficient n8 bytes)...}, wire-to-label
    section {u64 per wire}; here one constraint (3*w2 + 1) * w3 == w1 - w2 over wires {one, out, a, b}, with the
    sections in the order circom writes them."""
    u32 = lambda v: bytes([(v >> (8 * i)) & 0xFF for i in range(4)])  # noqa: E731
    u64 = lambda v: bytes([(v >> (8 * i)) & 0xFF for i in range(8)])  # noqa: E731
    hdr = u32(32) + P_LE + u32(4) + u32(1) + u32(0) + u32(2) + u64(4) + u32(1)
    lc_a = u32(2) + u32(0) + _fe(1) + u32(2) + _fe(3)
    lc_b = u32(1) + u32(3) + _fe(1)
    lc_c = u32(2) + u32(1) + _fe(1) + u32(2) + _fe(F.P - 1)
    cons = lc_a + lc_b + lc_c
    w2l = u64(0) + u64(1) + u64(2) + u64(3)
    data = (b"r1cs" + u32(1) + u32(3)
            + u32(1) + u64(len(hdr)) + hdr + u32(2) + u64(len(cons)) + cons + u32(3) + u64(len(w2l)) + w2l)
    r = X.read_r1cs(data)
    assert (r["prime"], r["n_wires"], r["n_pub_out"], r["n_pub_in"], r["n_prv_in"], r["n_labels"]) == (F.P, 4, 1, 0, 2, 4)
    assert r["constraints"] == [({0: 1, 2: 3}, {3: 1}, {1: 1, 2: F.P - 1})] and r["wire2label"] == [0, 1, 2, 3]
    assert X.write_r1cs(4, 1, 0, 2, r["constraints"]) == data
    a, b = 12345, F.P - 7
    good = [1, ((3 * a + 1) * b + a) % F.P, a, b]
    assert X.failed_constraints(r, good) == []
    assert X.failed_constraints(r, good[:1] + [good[1] + 1] + good[2:]) == [0]
    # constraint section before the header (readers must not assume an order)
    moved = (b"r1cs" + u32(1) + u32(3)
             + u32(2) + u64(len(cons)) + cons + u32(3) + u64(len(w2l)) + w2l + u32(1) + u64(len(hdr)) + hdr)
    assert X.read_r1cs(moved) == r


def test_oracle_header_symbol_lives_in_the_checker_library_only():
    import ctypes
    import re

    from oracle import c_oracle
    from zkir_amd import api
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "zkir_oracle.h")).read(), flags=re.S)
    names = set(re.findall(r"\b(zkir_[a-z0-9_]+)\s*\(", hdr))
    assert names == {"zkir_eval_cpu"}
    c_oracle.build()
    checker = ctypes.CDLL(os.path.join(root, "oracle", "_build", "libzkir_oracle.so"))
    product = ctypes.CDLL(api.LIB_PATH)
    assert hasattr(checker, "zkir_eval_cpu") and not hasattr(product, "zkir_eval_cpu")
