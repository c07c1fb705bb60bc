"""Host side of the library (no GPU): the C++ mirror of the reference's rng /
generator / circom against the committed fixtures and the Python restatement, the
s-expression form, lower against the direct AST interpreter, and the C ABI's
error behaviour."""
import ctypes
import json
import os
import random
import re

import pytest

import zkir_amd as z
from oracle import c_oracle as C
from oracle import fr_ref as F
from oracle import zkir_ref as R
from zkir_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def ext_config(**kw):
    g = z.default_generator_config()
    for k, v in kw.items():
        setattr(g, k, v)
    return g


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "zkir_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(zkir_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 29
    lib = ctypes.CDLL(api.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_struct_sizes_match_header():
    assert ctypes.sizeof(api.TapeHdr) == 88
    assert ctypes.sizeof(api.GeneratorConfig) == 20 * 4
    assert ctypes.sizeof(api.LowerOptions) == 14 * 4


def test_default_config_is_the_reference_literals():
    g = z.default_generator_config()
    # src/generator.rs:147, :167, :224, :486, :406, :416, :436, :264
    assert (g.min_instructions, g.max_instructions, g.statement_budget, g.max_array_len, g.max_expr_depth,
            g.loop_init_max, g.loop_bound_max, g.loop_body_max, g.name_len) == (15, 30, 10, 10, 1, 30, 60, 5, 8)
    assert (g.inputs_readable, g.arithmetic_only, g.quadratic_aware, g.forced_inputs, g.no_loops,
            g.safe_array_index) == (0,) * 6


def test_rng_stream_against_fixture():
    for seed, ent in load("rng_kat.json").items():
        assert [hex(x) for x in z.rng_stream(int(seed, 16), 16)] == ent["next"]
    with pytest.raises(z.ZkirError):
        z.rng_stream(0, 4)  # seed 0 is the xorshift's fixed point


def test_generator_and_transpiler_against_fixture():
    gold = load("circuits.json")
    for seed, ent in gold["reference"].items():
        c = z.Generator(int(seed, 16)).generate()
        assert c.text() == ent["text"]
        assert z.CircomTranspiler(c).transpile_circuit() == ent["circom"]
    for seed, ent in gold["extended"].items():
        c = z.Generator(int(seed, 16), ext_config(**ent["config"])).generate()
        assert c.text() == ent["text"]
        assert z.CircomTranspiler(c).transpile_circuit() == ent["circom"]


def test_generator_matches_restatement_over_seeds():
    for seed in list(range(1, 120)) + [2**63, 2**64 - 1]:
        assert z.Generator(seed).generate().text() == R.print_text(R.Generator(R.Rng(seed)).generate())
    cfg = dict(min_instructions=200, max_instructions=200, forced_inputs=5, inputs_readable=1, arithmetic_only=1,
               quadratic_aware=1, no_loops=1, safe_array_index=1, signal_window=6)
    for seed in range(1, 12):
        assert z.Generator(seed, ext_config(**cfg)).generate().text() == \
            R.print_text(R.Generator(R.Rng(seed), **cfg).generate())
    cfg = dict(cfg, max_linear_terms=1, signal_window=0)
    for seed in range(1, 8):
        assert z.Generator(seed, ext_config(**cfg)).generate().text() == \
            R.print_text(R.Generator(R.Rng(seed), **cfg).generate())
    # every factor of a constraint a signal: the constant-or-signal draw of a factor is not made
    for terms in (1, 2):
        cfg2 = dict(cfg, signal_factors=1, max_linear_terms=terms)
        for seed in range(1, 8):
            text = z.Generator(seed, ext_config(**cfg2)).generate().text()
            assert text == R.print_text(R.Generator(R.Rng(seed), **cfg2).generate())
            assert text != z.Generator(seed, ext_config(**dict(cfg2, signal_factors=0))).generate().text()
    # hundreds of written arrays: gen_array_index's one draw per array goes through the generator's jump tables
    cfg = dict(cfg, min_instructions=1500, max_instructions=1500)
    for seed in (3, 4):
        text = z.Generator(seed, ext_config(**cfg)).generate().text()
        assert text == R.print_text(R.Generator(R.Rng(seed), **cfg).generate())
        assert text.count("(array ") > 100
    cfg = dict(min_instructions=40, max_instructions=80, max_expr_depth=3, inputs_readable=1, safe_array_index=1)
    for seed in range(1, 40):
        assert z.Generator(seed, ext_config(**cfg)).generate().text() == \
            R.print_text(R.Generator(R.Rng(seed), **cfg).generate())
    # If statements (the reference keeps them switched off): as written, and with decidable conditions
    n_if = 0
    for mode in (1, 2):
        cfg = dict(enable_if=mode, inputs_readable=1, safe_array_index=1)
        for seed in range(1, 60):
            text = z.Generator(seed, ext_config(**cfg)).generate().text()
            assert text == R.print_text(R.Generator(R.Rng(seed), **cfg).generate())
            n_if += text.count("(if ")
            assert "(if " not in z.Generator(seed).generate().text()
    assert n_if > 50


def test_transpiler_tokens_one_per_operator():
    """Each Expr variant prints as the Circom token of src/circom.rs:35-157, fully parenthesised."""
    want = {"add": "+", "sub": "-", "mul": "*", "pow": "**", "div": "/", "idiv": "\\", "rem": "%", "lt": "<",
            "le": "<=", "gt": ">", "ge": ">=", "eq": "==", "and": "&&", "or": "||", "band": "&", "bor": "|",
            "bxor": "^", "shr": ">>", "shl": "<<"}
    for op, tok in want.items():
        txt = '(circuit (vars (var 0 "x" field local (init 0))) (instrs (decl 0 (%s (const "1") (const "2")))))' % op
        out = z.CircomTranspiler(z.Circuit.parse(txt)).transpile_circuit()
        assert out == "\ntemplate Program() {\n   var x = (1%s2);\n}\n\ncomponent main = Program();\n        " % tok
    txt = '(circuit (vars (var 0 "x" field local (init 0))) (instrs (decl 0 (bnot (const "9")))))'
    assert "   var x = (~9);\n" in z.CircomTranspiler(z.Circuit.parse(txt)).transpile_circuit()


def test_transpiler_statements_by_hand():
    """if / else, while, for, arrays, signals: the text of src/circom.rs:161-410, written out by hand."""
    txt = '''(circuit (vars (var 0 "i" field local (init 0)) (var 1 "a" (array 4) local (init 0 0 0 0))
        (var 2 "inp" (array 2) input (init 0 0)) (var 3 "o" field output (init 0)) (var 4 "w" field local (init 0)))
      (instrs (decl 1) (decl 2) (decl 3) (decl 4)
        (for (decl 0 (const "3")) (le (var 0) (const "5")) (assign 0 (add (var 0) (const "1")))
             (body (aassign 1 (const "2") (var (2 1)))
                   (if (lt (var 0) (const "4")) (then (assign 4 (const "7"))) (else (assign 4 (const "8")))) ))
        (while (lt (var 4) (const "9")) (body (assign 4 (add (var 4) (const "1")))))
        (if (eq (var 4) (const "9")) (then (constrain 3 (var 4))))))'''
    want = ("\ntemplate Program() {\n"
            "   var a[4];\n"
            "   signal input inp[2];\n"
            "   signal output o;\n"
            "   var w;\n"
            "   for (var i = 3; (i<=5); i=(i+1)) {\n"
            "      a[2]=(inp[1]);\n"
            "      if ((i<4)) {\n"
            "         w=7;\n"
            "      } else {\n"
            "         w=8;\n"
            "      }\n"
            "   }\n"
            "   while ((w<9)) {\n"
            "      w=(w+1);\n"
            "   }\n"
            "   if ((w==9)) {\n"
            "      o<==w;\n"
            "   }\n"
            "}\n\ncomponent main = Program();\n        ")
    c = z.Circuit.parse(txt)
    assert z.CircomTranspiler(c).transpile_circuit() == want
    assert R.transpile_circuit(R.parse_text(txt)) == want
    # the asserts of circom.rs:166-175, :203-207 become errors, not aborts
    bad = '(circuit (vars (var 0 "s" field input (init 0))) (instrs (decl 0) (assign 0 (const "1"))))'
    with pytest.raises(z.ZkirError):
        z.CircomTranspiler(z.Circuit.parse(bad)).transpile_circuit()


def test_debug_print_is_rust_pretty_debug():
    txt = '(circuit (vars (var 0 "v" field local (init 0))) (instrs (decl 0 (add (const "1") (var 0))) (assign 0 (var 0))))'
    want = """[
    VarDecl(
        Variable {
            var_ref: VarRef {
                id: 0,
                idx: None,
            },
            name: "v",
            var_type: Field,
            role: Local,
            initialized: [
                false,
            ],
        },
        Some(
            Add(
                Constant(
                    "1",
                ),
                Var(
                    VarRef {
                        id: 0,
                        idx: None,
                    },
                ),
            ),
        ),
    ),
    Assign(
        VarRef {
            id: 0,
            idx: None,
        },
        Var(
            VarRef {
                id: 0,
                idx: None,
            },
        ),
    ),
]"""
    assert z.Circuit.parse(txt).debug() == want


def test_text_form_round_trip_and_errors():
    for seed in range(1, 30):
        t = z.Generator(seed).generate().text()
        assert z.Circuit.parse(t).text() == t
        assert R.print_text(R.parse_text(t)) == t
    for bad in ["", "(", "(circuit)", "(circuit (vars) (instrs (decl 3)))", "(circuit (vars) (instrs (frob 1)))",
                '(circuit (vars (var 0 "a" field local (init 0)) (var 0 "b" field local (init 0))) (instrs))',
                "(circuit (vars) (instrs)) trailing", '(circuit (vars) (instrs (assign 0 (const 5))))']:
        with pytest.raises(z.ZkirError) as ei:
            z.Circuit.parse(bad)
        assert ei.value.code == api.E_INVALID


def classify_and_values(circ_text, ref, inputs, fold, bnot=0, cap=2000):
    o = z.default_lower_options()
    o.fold_constants, o.max_iterations, o.bnot_variant = fold, cap, bnot
    try:
        tape, cls = z.Lowerer(z.Circuit.parse(circ_text), o).lower(), "OK"
    except z.LowerRejected as e:
        tape, cls = None, api.LOWER_CLASSES[e.status_class]
    try:
        it, rcls = R.Interpreter(ref, inputs, bnot_variant=bnot, max_iterations=cap).run(), "OK"
    except R.Rejected as e:
        it, rcls = None, e.cls
    return tape, cls, it, rcls


def n_inputs_of(ref):
    return sum((v["array_len"] or 1) for v in ref["vars"] if v["role"] == "input")


def test_lower_matches_ast_interpreter_reference_shaped():
    """lower + tape semantics == direct big-integer interpretation of the AST, and the
    same circuits are refused for the same reason."""
    seen = set()
    for seed in range(1, 90):
        ref = R.Generator(R.Rng(seed)).generate()
        text = R.print_text(ref)
        rnd = random.Random(seed)
        inputs = [rnd.randrange(F.P) for _ in range(n_inputs_of(ref))]
        for fold in (1, 0):
            tape, cls, it, rcls = classify_and_values(text, ref, inputs, fold, cap=300)
            assert cls == rcls, (seed, fold)
            seen.add(cls)
            if tape is None:
                continue
            outs, unsat, faults = F.eval_tape_one(tape.ops(), tape.consts(), inputs)
            assert [outs[k] for k in range(tape.hdr.n_outputs)] == it.outputs
            assert unsat == it.unsat() == [] and tape.hdr.n_constraints == len(it.constraints)
            assert bool(faults) == bool(it.faults)
    assert {"OK", "LoopCap", "DoubleAssign"} <= seen


EXT = dict(min_instructions=120, max_instructions=160, forced_inputs=6, inputs_readable=1, safe_array_index=1)


def test_lower_matches_ast_interpreter_extended():
    seen = set()
    for seed in range(1, 60):
        cfg = dict(EXT, max_expr_depth=1 + seed % 3, arithmetic_only=seed % 2, quadratic_aware=(seed // 2) % 2,
                   no_loops=(seed // 4) % 2, enable_if=(seed // 8) % 3)
        ref = R.Generator(R.Rng(seed), **cfg).generate()
        text = R.print_text(ref)
        assert z.Generator(seed, ext_config(**cfg)).generate().text() == text
        rnd = random.Random(seed)
        special = [0, 1, F.P - 1, F.HALF, F.HALF + 1, 2, 253, 254]
        inputs = [rnd.choice(special) if rnd.random() < 0.3 else rnd.randrange(F.P) for _ in range(n_inputs_of(ref))]
        for fold, bnot in ((1, 0), (0, 1)):
            tape, cls, it, rcls = classify_and_values(text, ref, inputs, fold, bnot, cap=400)
            assert cls == rcls, (seed, fold, cls, rcls)
            seen.add(cls)
            if tape is None:
                continue
            outs, unsat, faults = F.eval_tape_one(tape.ops(), tape.consts(), inputs)
            assert [outs[k] for k in range(tape.hdr.n_outputs)] == it.outputs, seed
            assert unsat == it.unsat() == []
            assert tape.hdr.n_constraints == len(it.constraints)
            # the triples table repeats the operands of the CHECK ops, in order
            # (a reordered tape's CHECK ops carry their constraint number + 1 above the opcode)
            checks = [((op[0] >> 8) - 1 if op[0] >> 8 else k, (op[2], op[3], op[1]))
                      for k, op in enumerate(o for o in tape.ops() if o[0] & 0xFF == F.OP_CHECK)]
            assert [c for _, c in sorted(checks)] == tape.triples()
            # the C interpreter agrees with the big-integer one on the same tape
            n = 3
            vecs = [inputs] + [[rnd.randrange(F.P) for _ in inputs] for _ in range(n - 1)]
            hdr = bytes(tape.hdr)
            o_c, u_c, first_c, f_c = C.eval_cpu(hdr, tape.blob(), z.pack_inputs(vecs, tape.hdr.n_inputs), n,
                                                tape.hdr.n_outputs)
            assert z.unpack_outputs(o_c)[0] == it.outputs and u_c == 0 and first_c is None
    assert {"OK", "UnknownControl"} <= seen, seen


def test_lower_classes_by_hand():
    base = '''(circuit (vars (var 0 "a" field input (init 0)) (var 1 "b" field input (init 0))
      (var 2 "o" field output (init 0)) (var 3 "t" field local (init 0)) (var 4 "arr" (array 3) local (init 0 0 0))
      (var 5 "oo" (array 2) output (init 0 0)))
     (instrs (decl 0) (decl 1) (decl 2) (decl 3) (decl 4) (decl 5) %s))'''
    cases = {
        "OK": '(constrain 2 (add (mul (add (var 0) (const "1")) (var 1)) (const "3")))',
        "NonQuadratic": '(constrain 2 (lt (var 0) (var 1)))',
        "DoubleAssign": '(for (decl 3 (const "0")) (le (var 3) (const "1")) (assign 3 (add (var 3) (const "1"))) (body (constrain 2 (var 0))))',
        "OOBIndex": '(aassign 4 (const "3") (const "1"))',
        "DivByZero": '(assign 3 (rem (const "4") (const "0")))',
        "UnknownControl": '(aassign 4 (var 0) (const "1"))',
        "UninitSignal": '(assign 3 (var (5 1)))',
        "LoopCap": '(for (decl 3 (const "0")) (le (var 3) (const "5")) (assign 3 (add (var 3) (const "1"))) (body (assign 3 (const "0"))))',
        "BadAst": '(assign 9 (const "1"))',
    }
    for want, stmt in cases.items():
        text = base % stmt
        ref = R.parse_text(text)
        tape, cls, it, rcls = classify_and_values(text, ref, [5, 6], 1, cap=500)
        assert cls == want and rcls == want, (want, cls, rcls)
    # a signal constrained to a constant is still a signal: its value decides neither a condition nor an index
    for stmt in ('(constrain 2 (const "5")) (if (eq (var 2) (const "5")) (then (assign 3 (const "1"))))',
                 '(constrain 2 (const "1")) (aassign 4 (var 2) (const "7"))',
                 '(constrain 2 (const "2")) (assign 3 (add (var 2) (const "1"))) (while (lt (var 3) (const "2")) (body (assign 3 (const "9"))))'):
        text = base % stmt
        tape, cls, it, rcls = classify_and_values(text, R.parse_text(text), [5, 6], 1)
        assert cls == rcls == "UnknownControl", (stmt, cls, rcls)
    # ... while arithmetic on it is still folded where that is allowed
    text = base % '(constrain 2 (const "5")) (constrain (5 0) (mul (var 2) (var 0)))'
    tape, cls, it, rcls = classify_and_values(text, R.parse_text(text), [5, 6], 1)
    assert cls == rcls == "OK" and it.outputs[1] == 25
    # division by a known constant keeps the degree (x / k == x * k^-1)
    text = base % '(constrain 2 (div (mul (var 0) (var 1)) (const "7")))'
    tape, cls, it, rcls = classify_and_values(text, R.parse_text(text), [5, 6], 1)
    assert cls == rcls == "OK"
    outs, unsat, _ = F.eval_tape_one(tape.ops(), tape.consts(), [5, 6])
    assert outs[0] == F.div(30, 7) == it.outputs[0] and unsat == [] and outs[1] == outs[2] == 0


def test_slot_allocation_options_do_not_change_results():
    cfg = dict(EXT, arithmetic_only=1, quadratic_aware=1, no_loops=1, min_instructions=300, max_instructions=300)
    c = z.Generator(5, ext_config(**cfg)).generate()
    rnd = random.Random(1)
    base = None
    for nl, reuse, acc in [(0, 0, 0), (0, 1, 0), (4, 1, 1), (8, 1, 1), (16, 1, 1), (16, 0, 1), (1, 1, 1)]:
        o = z.default_lower_options()
        o.n_lslots, o.reuse_slots, o.use_acc = nl, reuse, acc
        t = z.Lowerer(c, o).lower()
        assert t.hdr.n_lslots <= nl
        if base is None:
            inputs = [rnd.randrange(F.P) for _ in range(t.hdr.n_inputs)]
        res = F.eval_tape_one(t.ops(), t.consts(), inputs)
        base = base or res
        assert res == base
        if not reuse and not nl:
            assert t.hdr.n_gslots >= 250  # one slot per stored value
    assert z.Lowerer(c, None).lower().hdr.n_constraints > 20


def test_statement_order_and_window_policy_do_not_change_results():
    """A plain tape with its statements reordered, the ops of each statement reordered and its window run as a cache gives, vector for vector, the outputs,
    the unsatisfied pairs and the first unsatisfied pair of the tape in program order (C checker, Python interpreter);
    a broken constraint keeps its number; the reordered tape moves fewer values through device slots."""
    import struct
    import numpy as np
    sys_path_tools = os.path.join(ROOT, "tools")
    import sys
    sys.path.insert(0, sys_path_tools)
    from slot_traffic import slot_traffic
    rnd = random.Random(4)
    for seed, n_instr, terms in ((5, 300, 1), (9, 800, 1), (11, 60, 1), (13, 400, 2)):
        cfg = dict(EXT, arithmetic_only=1, quadratic_aware=1, no_loops=1, min_instructions=n_instr, max_instructions=n_instr,
                   max_linear_terms=terms)
        c = z.Generator(seed, ext_config(**cfg)).generate()
        tapes = {}
        for reorder, cache, within in ((0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 0, 1), (0, 1, 1), (1, 1, 1)):
            o = z.default_lower_options()
            o.reorder, o.window_cache, o.order_ops = reorder, cache, within
            tapes[reorder, cache, within] = z.Lowerer(c, o).lower()
        ref = tapes[0, 0, 0]
        assert not any(op[0] >> 8 for op in ref.ops()) and all(op[0] >> 8 for op in tapes[1, 1, 1].ops() if op[0] & 0xFF == F.OP_CHECK)
        vecs = [[rnd.randrange(F.P) for _ in range(ref.hdr.n_inputs)] for _ in range(9)]
        inp = z.pack_inputs(vecs, ref.hdr.n_inputs)

        def broken(t, number):  # the c operand of one constraint replaced by a constant: it fails for every vector
            ops, k = [list(op) for op in t.ops()], 0
            for op in ops:
                if op[0] & 0xFF == F.OP_CHECK:
                    if ((op[0] >> 8) - 1 if op[0] >> 8 else k) == number:
                        op[1] = F.operand(F.K_CONST, 0)
                    k += 1
            return b"".join(struct.pack("<4I", *op) for op in ops) + t.blob()[16 * t.hdr.n_ops:]

        n_con = ref.hdr.n_constraints
        want = C.eval_cpu(bytes(ref.hdr), ref.blob(), inp, len(vecs), ref.hdr.n_outputs)
        want_bad = C.eval_cpu(bytes(ref.hdr), broken(ref, n_con // 2), inp, len(vecs), ref.hdr.n_outputs)
        assert want[1] == 0 and want_bad[1] == len(vecs) and want_bad[2] == (0, n_con // 2)
        for key, t in tapes.items():
            assert (t.hdr.n_constraints, t.hdr.n_mul_ops, t.hdr.n_outputs) == (n_con, ref.hdr.n_mul_ops, ref.hdr.n_outputs)
            got = C.eval_cpu(bytes(t.hdr), t.blob(), inp, len(vecs), t.hdr.n_outputs)
            assert np.array_equal(got[0], want[0]) and got[1:] == want[1:], key
            got_bad = C.eval_cpu(bytes(t.hdr), broken(t, n_con // 2), inp, len(vecs), t.hdr.n_outputs)
            assert np.array_equal(got_bad[0], want_bad[0]) and got_bad[1:] == want_bad[1:], key
            assert F.eval_tape_one(t.ops(), t.consts(), vecs[0]) == F.eval_tape_one(ref.ops(), ref.consts(), vecs[0])
            z.Tape.from_parts(t.hdr, t.blob())  # validation accepts numbered CHECK ops in a plain tape
        if n_instr >= 300:
            r0, w0, _ = slot_traffic(tapes[0, 0, 0].ops())
            r1, w1, _ = slot_traffic(tapes[1, 1, 0].ops())
            assert r1 + w1 < r0 + w0, (r0, w0, r1, w1)
        if terms == 2:  # `(a+b)*(c+d) + (e+f)`: with the addend first the product is never stored
            r0, w0, _ = slot_traffic(tapes[0, 1, 0].ops())
            r1, w1, _ = slot_traffic(tapes[0, 1, 1].ops())
            assert r1 + w1 < 0.9 * (r0 + w0), (r0, w0, r1, w1)
    # numbers on some CHECK ops only: refused
    t = tapes[1, 1, 1]
    ops = [list(op) for op in t.ops()]
    k = next(i for i, op in enumerate(ops) if op[0] & 0xFF == F.OP_CHECK)
    ops[k][0] &= 0xFF
    with pytest.raises(z.ZkirError):
        z.Tape.from_parts(t.hdr, b"".join(struct.pack("<4I", *op) for op in ops) + t.blob()[16 * t.hdr.n_ops:])


def test_tape_validation_rejects_out_of_range_operands():
    c = z.Generator(7, ext_config(**dict(EXT, arithmetic_only=1, quadratic_aware=1, no_loops=1))).generate()
    t = z.Lowerer(c).lower()
    blob = bytearray(t.blob())
    z.Tape.from_parts(t.hdr, bytes(blob))  # the untouched tape is accepted
    import struct
    for field, value in [(1, F.operand(F.K_GSLOT, t.hdr.n_gslots)), (2, F.operand(F.K_CONST, t.hdr.n_consts)),
                         (2, F.operand(F.K_LSLOT, 16)), (0, 200)]:
        b = bytearray(blob)
        # corrupt the first binary value op
        k = next(i for i, op in enumerate(t.ops()) if op[0] == F.OP_ADD)
        struct.pack_into("<I", b, 16 * k + 4 * field, value)
        with pytest.raises(z.ZkirError):
            z.Tape.from_parts(t.hdr, bytes(b))
    with pytest.raises(z.ZkirError):
        z.Tape.from_parts(t.hdr, bytes(blob[:-4]))


def test_gadget_shape():
    c = z.poseidon_style_gadget(3, 5, 99)
    t = z.Lowerer(c).lower()
    assert t.hdr.n_constraints == 3 * 3 * 5 and t.hdr.n_inputs == 3 and t.hdr.n_outputs == 45
    ref = R.parse_text(c.text())
    it = R.Interpreter(ref, [1, 2, 3]).run()
    outs, unsat, _ = F.eval_tape_one(t.ops(), t.consts(), [1, 2, 3])
    assert [outs[k] for k in range(45)] == it.outputs and unsat == []
    assert "signal output sq_0_0;" in z.CircomTranspiler(c).transpile_circuit()


def test_cli_reproduces_reference_main(tmp_path):
    """`cargo run` with a fixed seed (src/main.rs:81-92): progress lines, {:#?}, source, and the file."""
    from zkir_amd import cli
    for seed in (3, 12345):
        g = R.Generator(R.Rng(seed))
        ref = g.generate()
        text, source = cli.render(seed)
        assert source == R.transpile_circuit(ref)
        lines = text.split("\n")
        assert lines[0] == "Generating %d instructions..." % g.requested
        assert lines[1] == "Generated %d/%d instructions" % (g.generated, g.requested)
        assert lines[2] == "[" and text.endswith(source + "\n")
        assert 15 <= g.requested <= 30 and g.generated <= g.requested
    out = tmp_path / "circuit.circom"
    cli.main(["--seed", "3", "--out", str(out)])
    assert out.read_text() == cli.render(3)[1]


def test_build_entry_does_not_import_the_package():
    """build() must work in a checkout that has no libzkir_hip.so yet: it may not import zkir_amd
    (whose import loads the library) before it has built it."""
    import subprocess
    import sys
    code = ("import sys, __graft_entry__ as g; m = g._build_module(); "
            "assert 'zkir_amd' not in sys.modules and 'zkir_amd.api' not in sys.modules; "
            "assert callable(m.build); print('ok')")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr


def test_reducer_refuses_bad_arguments_and_needs_a_gpu():
    """zkir_reducer_create: argument checks come before any device work; with no GPU it answers NODEVICE."""
    from zkir_amd import api
    for devices, counters, cap in (([], 4, 8), ([0, 0], 4, 8), ([0], 0, 8), ([0], 4, (1 << 20) + 1)):
        with pytest.raises(api.ZkirError) as e:
            api.Reducer(devices, counters, cap)
        assert e.value.code in (api.E_INVALID, api.E_NODEVICE)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(api.ZkirError) as e:
            api.Reducer([0], 4, 8)
        assert e.value.code == api.E_NODEVICE
