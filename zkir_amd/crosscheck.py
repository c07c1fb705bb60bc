"""Cross-check harness against iden3/circom, for a machine that has circom and snarkjs.

The reference's only backend prints Circom text (src/circom.rs:412-429) for circom to compile;
what this evaluator computes is meant to equal what circom's witness calculator computes for
that text.  Neither tool can run where this repository is built, so this module does the two
halves that need no circom:

  emit     circuit.circom + input_K.json per vector + expected_K.json (this evaluator's outputs)
           + run.sh with the circom / snarkjs commands;
  compare  read circom's witness_K.wtns and circuit.sym (and circuit.r1cs), and report every
           output signal that differs and every R1CS constraint the witness fails.

File formats follow iden3's published binary layouts (.wtns v2, .r1cs v1) and circom's --sym
text; they are restated here from the specifications and checked only against files written by
this module (no circom output exists in the reference to pin them on).
"""
import json
import os
import re
import struct

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


# ---- names -------------------------------------------------------------------
def variable_table(circuit):
    """{variable id: (name, element count, role)} from the circuit's text form."""
    out = {}
    for m in re.finditer(r'\(var (\d+) "([^"]*)" (field|\(array (\d+)\)) (input|output|local)', circuit.text()):
        out[int(m.group(1))] = (m.group(2), int(m.group(4)) if m.group(4) else 1, m.group(5), m.group(3) != "field")
    return out


def input_json(circuit, tape, vector):
    """circom's input.json for one input vector (values in tape input order)."""
    names = variable_table(circuit)
    doc = {}
    for var_id, first, count in tape.io_map(0):
        name, _, _, is_array = names[var_id]
        vals = [str(int(vector[first + k])) for k in range(count)]
        doc[name] = vals if is_array else vals[0]
    return doc


def expected_outputs(circuit, tape, outputs):
    """{output signal name: decimal or list} from one vector's outputs (tape output order)."""
    names = variable_table(circuit)
    doc = {}
    for var_id, first, count in tape.io_map(1):
        name, _, _, is_array = names[var_id]
        vals = [str(int(outputs[first + k])) for k in range(count)]
        doc[name] = vals if is_array else vals[0]
    return doc


RUN_SH = """#!/bin/sh
# needs circom (>= 2.0) and snarkjs on PATH; run inside this directory
circom circuit.circom --r1cs --wasm --sym > circom.log 2>&1 || { echo "circom refused the circuit (see circom.log)"; exit 0; }
set -e
for f in input_*.json; do
  k=${f#input_}; k=${k%.json}
  node circuit_js/generate_witness.js circuit_js/circuit.wasm "$f" "witness_$k.wtns"
  snarkjs wtns check circuit.r1cs "witness_$k.wtns"
done
"""


def emit_case(directory, circuit, circom_text, tape, vectors, outputs=None):
    """Write the files circom needs for `vectors` (and what this evaluator got, if given)."""
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, "circuit.circom"), "w") as f:
        f.write(circom_text)
    for k, vec in enumerate(vectors):
        with open(os.path.join(directory, "input_%d.json" % k), "w") as f:
            json.dump(input_json(circuit, tape, vec), f)
        if outputs is not None:
            with open(os.path.join(directory, "expected_%d.json" % k), "w") as f:
                json.dump(expected_outputs(circuit, tape, outputs[k]), f)
    path = os.path.join(directory, "run.sh")
    with open(path, "w") as f:
        f.write(RUN_SH)
    os.chmod(path, 0o755)


# ---- iden3 binary containers -----------------------------------------------------
def _sections(data, magic, versions):
    if data[:4] != magic:
        raise ValueError("not a %s file" % magic.decode())
    version, n = struct.unpack_from("<II", data, 4)
    if version not in versions:
        raise ValueError("%s version %d not supported" % (magic.decode(), version))
    pos, out = 12, {}
    for _ in range(n):
        if pos + 12 > len(data):
            raise ValueError("truncated section table")
        sid, size = struct.unpack_from("<IQ", data, pos)
        pos += 12
        if pos + size > len(data):
            raise ValueError("section %d runs past the end of the file" % sid)
        out.setdefault(sid, data[pos:pos + size])
        pos += size
    return out


def _container(magic, version, sections):
    out = magic + struct.pack("<II", version, len(sections))
    for sid, body in sections:
        out += struct.pack("<IQ", sid, len(body)) + body
    return out


def read_wtns(path_or_bytes):
    """-> (prime, [witness values]); wire 0 is the constant 1."""
    data = path_or_bytes if isinstance(path_or_bytes, bytes) else open(path_or_bytes, "rb").read()
    sec = _sections(data, b"wtns", (1, 2))
    n8, = struct.unpack_from("<I", sec[1], 0)
    prime = int.from_bytes(sec[1][4:4 + n8], "little")
    n_w, = struct.unpack_from("<I", sec[1], 4 + n8)
    body = sec[2]
    if len(body) != n_w * n8:
        raise ValueError("witness section holds %d bytes, header says %d values of %d" % (len(body), n_w, n8))
    return prime, [int.from_bytes(body[i * n8:(i + 1) * n8], "little") for i in range(n_w)]


def write_wtns(values, prime=P, n8=32):
    hdr = struct.pack("<I", n8) + prime.to_bytes(n8, "little") + struct.pack("<I", len(values))
    body = b"".join(int(v).to_bytes(n8, "little") for v in values)
    return _container(b"wtns", 2, [(1, hdr), (2, body)])


def read_r1cs(path_or_bytes):
    """-> dict(prime, n_wires, n_pub_out, n_pub_in, n_prv_in, n_labels, constraints, wire2label);
    a constraint is (A, B, C), each {wire: coefficient}, meaning (A.w)*(B.w) == C.w."""
    data = path_or_bytes if isinstance(path_or_bytes, bytes) else open(path_or_bytes, "rb").read()
    sec = _sections(data, b"r1cs", (1,))
    h = sec[1]
    n8, = struct.unpack_from("<I", h, 0)
    prime = int.from_bytes(h[4:4 + n8], "little")
    n_wires, n_pub_out, n_pub_in, n_prv_in, n_labels, m = struct.unpack_from("<IIIIQI", h, 4 + n8)
    body, pos, cons = sec[2], 0, []
    for _ in range(m):
        lcs = []
        for _ in range(3):
            n_terms, = struct.unpack_from("<I", body, pos)
            pos += 4
            lc = {}
            for _ in range(n_terms):
                wire, = struct.unpack_from("<I", body, pos)
                lc[wire] = int.from_bytes(body[pos + 4:pos + 4 + n8], "little")
                pos += 4 + n8
            lcs.append(lc)
        cons.append(tuple(lcs))
    if pos != len(body):
        raise ValueError("constraint section has %d trailing bytes" % (len(body) - pos))
    w2l = list(struct.unpack_from("<%dQ" % n_wires, sec[3], 0)) if 3 in sec else []
    return {"prime": prime, "n_wires": n_wires, "n_pub_out": n_pub_out, "n_pub_in": n_pub_in,
            "n_prv_in": n_prv_in, "n_labels": n_labels, "constraints": cons, "wire2label": w2l}


def write_r1cs(n_wires, n_pub_out, n_pub_in, n_prv_in, constraints, prime=P, n8=32):
    hdr = struct.pack("<I", n8) + prime.to_bytes(n8, "little")
    hdr += struct.pack("<IIIIQI", n_wires, n_pub_out, n_pub_in, n_prv_in, n_wires, len(constraints))
    body = b""
    for con in constraints:
        for lc in con:
            body += struct.pack("<I", len(lc))
            for wire in sorted(lc):
                body += struct.pack("<I", wire) + (lc[wire] % prime).to_bytes(n8, "little")
    w2l = struct.pack("<%dQ" % n_wires, *range(n_wires))
    return _container(b"r1cs", 1, [(1, hdr), (2, body), (3, w2l)])


def read_sym(path_or_text):
    """circom --sym: lines `label,wire,component,name` -> {name: wire} (wire -1: optimised away)."""
    text = open(path_or_text).read() if os.path.exists(path_or_text) else path_or_text
    out = {}
    for line in text.splitlines():
        parts = line.strip().split(",")
        if len(parts) >= 4:
            out[",".join(parts[3:])] = int(parts[1])
    return out


# ---- verdicts -------------------------------------------------------------------------
# Phrases of circom's compile errors, as recalled from its sources (unverified here), and the class
# `lower` gives the same circuit.  A circuit both sides refuse for the same reason is an agreement.
CIRCOM_ERRORS = [
    ("Non quadratic constraints are not allowed", "NonQuadratic"),
    ("signal already assigned", "DoubleAssign"),
    ("Signal already assigned", "DoubleAssign"),
    ("Out of bounds", "OOBIndex"),
    ("out of bounds", "OOBIndex"),
    ("Division by zero", "DivByZero"),
    ("division by zero", "DivByZero"),
    ("can be unknown during the constraint generation phase", "UnknownControl"),
    ("Non-quadratic constraint was detected statically", "NonQuadratic"),
    ("depends on the value of the condition", "UnknownControl"),
]


def classify_circom_log(text):
    """The class named by circom's compile log: "Ok" when it reports no error, one of lower's class
    names for a recognised error, "Other" for an error this table does not know."""
    for phrase, cls in CIRCOM_ERRORS:
        if phrase in text:
            return cls
    return "Other" if "error" in text.lower() else "Ok"


# ---- comparison --------------------------------------------------------------------
def failed_constraints(r1cs, witness):
    """Indices of the R1CS constraints the witness does not satisfy (big integers)."""
    p = r1cs["prime"]

    def dot(lc):
        return sum(c * witness[w] for w, c in lc.items()) % p
    return [k for k, (a, b, c) in enumerate(r1cs["constraints"]) if dot(a) * dot(b) % p != dot(c)]


def compare_outputs(sym, witness, expected, component="main"):
    """[(signal, expected, circom's value)] for every output that differs or is missing."""
    bad = []
    for name, want in expected.items():
        wants = want if isinstance(want, list) else [want]
        for k, w in enumerate(wants):
            key = "%s.%s" % (component, name) + ("[%d]" % k if isinstance(want, list) else "")
            wire = sym.get(key, -1)
            got = witness[wire] if 0 <= wire < len(witness) else None
            if got is None or got != int(w):
                bad.append((key, int(w), got))
    return bad


def compare_verdict(directory):
    """(this evaluator's class, circom's class) for the circuit in `directory`, after run.sh."""
    with open(os.path.join(directory, "verdict.json")) as f:
        ours = json.load(f)["class"]
    log = os.path.join(directory, "circom.log")
    theirs = classify_circom_log(open(log).read()) if os.path.exists(log) else None
    return ("Ok" if ours == "OK" else ours), theirs


def compare_case(directory):
    """After run.sh: {vector: {"outputs": [...], "constraints": [...]}} for the vectors that disagree
    (and {"verdict": (ours, circom's)} when the two sides classify the circuit differently)."""
    if os.path.exists(os.path.join(directory, "verdict.json")) and os.path.exists(os.path.join(directory, "circom.log")):
        ours, theirs = compare_verdict(directory)
        if ours != theirs:
            return {"verdict": [ours, theirs]}
        if ours != "Ok":
            return {}
    sym = read_sym(os.path.join(directory, "circuit.sym"))
    r1cs_path = os.path.join(directory, "circuit.r1cs")
    r1cs = read_r1cs(r1cs_path) if os.path.exists(r1cs_path) else None
    report = {}
    k = 0
    while os.path.exists(os.path.join(directory, "witness_%d.wtns" % k)):
        prime, wit = read_wtns(os.path.join(directory, "witness_%d.wtns" % k))
        if prime != P:
            raise ValueError("witness is over another field")
        entry = {}
        exp_path = os.path.join(directory, "expected_%d.json" % k)
        if os.path.exists(exp_path):
            with open(exp_path) as f:
                diff = compare_outputs(sym, wit, json.load(f))
            if diff:
                entry["outputs"] = diff
        if r1cs is not None:
            bad = failed_constraints(r1cs, wit)
            if bad:
                entry["constraints"] = bad
        if entry:
            report[k] = entry
        k += 1
    return report


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    e = sub.add_parser("emit", help="write circuit.circom, input_K.json, expected_K.json, run.sh")
    e.add_argument("--seed", type=int, required=True)
    e.add_argument("--vectors", type=int, default=4)
    e.add_argument("--statements", type=int, default=0, help="0: the reference's 15..30")
    e.add_argument("--out", required=True)
    e.add_argument("--no-evaluate", action="store_true", help="inputs only (no GPU here)")
    e.add_argument("--extended", action="store_true",
                   help="generator extensions on (readable inputs, rank-1 constraints, no loops): circuits circom "
                        "accepts; without it the reference's own settings, most of whose circuits both sides refuse")
    e.add_argument("--max-iterations", type=int, default=4096, help="loop iterations lower unrolls before LoopCap")
    c = sub.add_parser("compare", help="read circom's files in a directory written by emit")
    c.add_argument("dir")
    args = ap.parse_args(argv)
    if args.cmd == "compare":
        rep = compare_case(args.dir)
        print(json.dumps(rep, indent=1))
        return 1 if rep else 0

    import random

    from . import api
    g = api.default_generator_config()
    if args.statements:
        g.min_instructions = g.max_instructions = args.statements
    g.inputs_readable = 1
    if args.extended:
        g.forced_inputs, g.arithmetic_only, g.quadratic_aware, g.no_loops, g.safe_array_index = 4, 1, 1, 1, 1
    circuit = api.Generator(args.seed, g).generate()
    text = api.CircomTranspiler(circuit).transpile_circuit()
    lo = api.default_lower_options()
    lo.max_iterations = args.max_iterations
    try:
        tape = api.Lowerer(circuit, lo).lower()
    except api.LowerRejected as e:
        # circom is expected to refuse this circuit too: record why, so that "both reject" can
        # be counted as agreement
        os.makedirs(args.out, exist_ok=True)
        with open(os.path.join(args.out, "circuit.circom"), "w") as f:
            f.write(text)
        with open(os.path.join(args.out, "verdict.json"), "w") as f:
            json.dump({"accepted": False, "class": api.LOWER_CLASSES[e.status_class], "message": str(e)}, f)
        print("wrote", args.out, "(rejected: %s)" % api.LOWER_CLASSES[e.status_class])
        return 0
    rnd = random.Random(args.seed)
    vectors = [[rnd.randrange(P) for _ in range(tape.hdr.n_inputs)] for _ in range(args.vectors)]
    outputs = None
    if not args.no_evaluate:
        ctx = api.Context(0)  # raises without an MI355X: nothing here falls back to the CPU
        tid = ctx.upload(tape)
        out, _, _, _ = ctx.eval_batch(tid, api.pack_inputs(vectors, tape.hdr.n_inputs), len(vectors))
        outputs = api.unpack_outputs(out)
        ctx.release(tid)
        ctx.close()
    emit_case(args.out, circuit, text, tape, vectors, outputs)
    with open(os.path.join(args.out, "verdict.json"), "w") as f:
        json.dump({"accepted": True, "class": "Ok", "n_constraints": tape.hdr.n_constraints}, f)
    print("wrote", args.out)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
