#!/usr/bin/env python3
"""Writes the cross-check bundles of this directory: for each case a circuit.circom (the text src/circom.rs prints),
input_K.json per vector, expected_K.json (what THIS evaluator computed on an MI355X, through the C ABI), verdict.json
(accepted, or the class lower refused the circuit with) and run.sh.  On a machine that has circom and snarkjs:

    cd tests/golden/crosscheck/<case> && ./run.sh && cd - && python -m zkir_amd.crosscheck compare tests/golden/crosscheck/<case>

prints {} when circom's witness calculator, its R1CS and its accept/refuse verdict agree with the files here, and
lists every output, constraint or verdict that differs otherwise: that run pins this repository's operator
semantics to circom's (nothing in the reference can: it holds no evaluator, no tests and no vectors).

Cases: the reference's own generator settings (constant right-hand sides; most seeds are refused, by circom too if
the classes of lower are right: the first four accepted seeds and one seed per refusal class are kept), the extended generator (rank-1 constraints over readable inputs), deep expressions over every operator,
and one hand-written circuit per operator whose outputs are (c1 op c2) * x for edge-case constants: those pin the
compile-time meaning of each Circom token of src/circom.rs:35-157, including both readings of `~`
(expected_K.json holds the 254-bit reading, expected_K.bnot256.json the 256-bit one).  An operator over two
constants is all the reference's AST can show circom of an operator's meaning: `<==` takes only quadratic
right-hand sides, and the AST has no `<--`.  The expected values of those cases were computed twice on the GPU,
folded on the host and, with folding off, by the every-operator kernel at run time: the two agree.

usage (needs an MI355X):  python tests/golden/crosscheck/make_bundles.py [out_dir]
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
import zkir_amd as z  # noqa: E402
from zkir_amd import api, crosscheck  # noqa: E402

P = crosscheck.P
HALF = P // 2
EDGE_PAIRS = [(7, 3), (P - 1, 2), (HALF + 1, 3), (HALF, HALF + 1), ((1 << 64) - 1, 255), (5, 300), (1 << 253, 1),
              (P - 2, P - 1), (12345678901234567890, P - 3), (0, 5), (1, 253), (P - 1, 254)]
OPS = ["add", "sub", "mul", "pow", "div", "idiv", "rem", "lt", "le", "gt", "ge", "eq", "and", "or", "band", "bor", "bxor",
       "shr", "shl"]


def operator_circuit(op):
    """one input x; out_k <== (c1 op c2) * x + 0 for every edge pair (unary ~: ~c1)"""
    pairs = [(a, b) for a, b in EDGE_PAIRS if not (op in ("div", "idiv", "rem") and b % P == 0)]
    if op == "pow":
        pairs = [(a, b) for a, b in pairs if b < (1 << 70)] + [(3, P - 2)]
    vars_ = ['(var 0 "x" field input (init 1))'] + ['(var %d "out%d" field output (init 0))' % (k + 1, k) for k in range(len(pairs))]
    instrs = ["(decl 0)"]
    for k, (a, b) in enumerate(pairs):
        e = '(bnot (const "%d"))' % a if op == "bnot" else '(%s (const "%d") (const "%d"))' % (op, a, b)
        instrs += ["(decl %d)" % (k + 1), '(constrain %d (add (mul %s (var 0)) (const "0")))' % (k + 1, e)]
    return "(circuit (vars %s) (instrs %s))" % (" ".join(vars_), " ".join(instrs))


def emit(ctx, out_dir, name, circuit, n_vectors, seed, bnot_variants=(0,), max_iterations=4096, also_unfolded=False):
    d = os.path.join(out_dir, name)
    text = api.CircomTranspiler(circuit).transpile_circuit()
    tapes = []
    for variant in bnot_variants:
        lo = api.default_lower_options()
        lo.max_iterations = max_iterations
        lo.bnot_variant = variant
        try:
            tapes.append(api.Lowerer(circuit, lo).lower())
        except api.LowerRejected as e:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "circuit.circom"), "w") as f:
                f.write(text)
            with open(os.path.join(d, "verdict.json"), "w") as f:
                json.dump({"accepted": False, "class": api.LOWER_CLASSES[e.status_class], "message": str(e)}, f)
            with open(os.path.join(d, "run.sh"), "w") as f:
                f.write(crosscheck.RUN_SH)
            os.chmod(os.path.join(d, "run.sh"), 0o755)
            return "refused (%s)" % api.LOWER_CLASSES[e.status_class]
    tape = tapes[0]
    rnd = random.Random(seed)
    vectors = [[rnd.choice([0, 1, P - 1, HALF, HALF + 1]) if rnd.random() < 0.15 else rnd.randrange(P) for _ in range(tape.hdr.n_inputs)]
               for _ in range(n_vectors)]
    results = []
    for variant, t in zip(bnot_variants, tapes):
        tid = ctx.upload(t)
        out, unsat, _, faults = ctx.eval_batch(tid, api.pack_inputs(vectors, t.hdr.n_inputs), len(vectors))
        ctx.release(tid)
        assert unsat == 0 and faults == 0, (name, unsat, faults)
        results.append(api.unpack_outputs(out))
        if also_unfolded:
            # the same circuit with nothing folded on the host: the operator over its two constants runs on the GPU
            # (the every-operator kernel) and must give what the host's folding gave
            lo = api.default_lower_options()
            lo.fold_constants, lo.bnot_variant = 0, variant
            t2 = api.Lowerer(circuit, lo).lower()
            tid = ctx.upload(t2)
            out2, unsat2, _, faults2 = ctx.eval_batch(tid, api.pack_inputs(vectors, t2.hdr.n_inputs), len(vectors))
            ctx.release(tid)
            assert (unsat2, faults2) == (0, 0) and api.unpack_outputs(out2) == results[-1], name
    crosscheck.emit_case(d, circuit, text, tape, vectors, results[0])
    if len(results) > 1:
        for k in range(n_vectors):
            with open(os.path.join(d, "expected_%d.bnot256.json" % k), "w") as f:
                json.dump(crosscheck.expected_outputs(circuit, tape, results[1][k]), f)
    with open(os.path.join(d, "verdict.json"), "w") as f:
        json.dump({"accepted": True, "class": "Ok", "n_constraints": tape.hdr.n_constraints}, f)
    return "ok (%d constraints)" % tape.hdr.n_constraints


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else HERE
    ctx = api.Context(0)  # raises without an MI355X: the expected values come from the product, not from a CPU stand-in
    log = {}
    # the reference's own settings (src/generator.rs literals): the first four seeds lower accepts, and the first
    # seed of every class it refuses with
    n_ok, classes = 0, set()
    for seed in range(1, 400):
        c = z.Generator(seed).generate()
        scan = api.default_lower_options()
        scan.max_iterations = 4096  # (a loop whose variable its body overwrites may never end: LoopCap, soon)
        try:
            api.Lowerer(c, scan).lower()
            cls = "Ok"
        except api.LowerRejected as e:
            cls = api.LOWER_CLASSES[e.status_class]
        if (cls == "Ok" and n_ok >= 4) or (cls != "Ok" and cls in classes):
            continue
        n_ok += cls == "Ok"
        classes.add(cls)
        log["reference_seed%d" % seed] = emit(ctx, out_dir, "reference_seed%d" % seed, c, 2, seed)
    for seed in range(1, 7):
        g = api.default_generator_config()
        g.min_instructions = g.max_instructions = 60
        g.forced_inputs, g.inputs_readable, g.arithmetic_only, g.quadratic_aware, g.no_loops, g.safe_array_index = 4, 1, 1, 1, seed % 2, 1
        log["extended_seed%d" % seed] = emit(ctx, out_dir, "extended_seed%d" % seed, z.Generator(seed, g).generate(), 3, seed, max_iterations=300)
    for seed in range(1, 5):
        g = api.default_generator_config()
        g.min_instructions, g.max_instructions, g.max_expr_depth = 30, 40, 3
        g.inputs_readable, g.safe_array_index, g.no_loops, g.quadratic_aware, g.forced_inputs = 1, 1, 1, 1, 3
        log["deep_seed%d" % seed] = emit(ctx, out_dir, "deep_seed%d" % seed, z.Generator(seed, g).generate(), 3, seed)
    for op in OPS + ["bnot"]:
        c = z.Circuit.parse(operator_circuit(op))
        log["operator_" + op] = emit(ctx, out_dir, "operator_" + op, c, 2, 77, bnot_variants=(0, 1) if op == "bnot" else (0,),
                                     also_unfolded=True)
    ctx.close()
    with open(os.path.join(out_dir, "index.json"), "w") as f:
        json.dump(log, f, indent=1, sort_keys=True)
    print(json.dumps(log, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
