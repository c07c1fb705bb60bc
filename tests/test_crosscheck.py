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
