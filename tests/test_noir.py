"""The Noir backend (second backend named by the reference's README.md:5,10; it has no code
there).  The expected text is written out by hand; no Noir compiler is available to try it on,
so what these tests pin is the mapping described in zkir_amd/csrc/host/noir.hpp."""
import pytest

import zkir_amd as z
from zkir_amd import api


def test_statements_by_hand():
    txt = '''(circuit (vars (var 0 "i" field local (init 0)) (var 1 "a" (array 4) local (init 0 0 0 0))
        (var 2 "inp" (array 2) input (init 0 0)) (var 3 "o" field output (init 0)) (var 4 "w" field local (init 0))
        (var 5 "x" field input (init 0)) (var 6 "oo" (array 2) output (init 0 0)))
      (instrs (decl 1) (decl 2) (decl 5) (decl 3) (decl 6) (decl 4 (const "5"))
        (for (decl 0 (const "3")) (le (var 0) (const "5")) (assign 0 (add (var 0) (const "1")))
             (body (aassign 1 (const "2") (var (2 1)))
                   (if (eq (var 0) (const "4")) (then (assign 4 (const "7"))) (else (assign 4 (mul (var 4) (var 0)))))))
        (if (sub (var 4) (const "9")) (then (constrain 3 (div (var 4) (var 5)))))
        (constrain (6 1) (add (mul (var 5) (var 5)) (var (1 2))))))'''
    want = ("fn main(inp: [Field; 2], x: Field) -> pub (Field, [Field; 2]) {\n"
            "    let mut o: Field = 0;\n"
            "    let mut oo: [Field; 2] = [0; 2];\n"
            "    let mut a: [Field; 4] = [0; 4];\n"
            "    let mut w: Field = 5;\n"
            "    for i_i in 3..(5 + 1) {\n"
            "        let i: Field = i_i as Field;\n"
            "        a[2] = inp[1];\n"
            "        if ((i == 4) as Field) != 0 {\n"
            "            w = 7;\n"
            "        } else {\n"
            "            w = (w * i);\n"
            "        }\n"
            "    }\n"
            "    if (w - 9) != 0 {\n"
            "        o = (w / x);\n"
            "    }\n"
            "    oo[1] = ((x * x) + a[2]);\n"
            "    (o, oo)\n"
            "}\n")
    assert z.NoirTranspiler(z.Circuit.parse(txt)).transpile_circuit() == want


def test_one_output_and_none():
    one = '(circuit (vars (var 0 "s" field input (init 0)) (var 1 "o" field output (init 0))) (instrs (decl 0) (decl 1) (constrain 1 (mul (var 0) (var 0)))))'
    assert z.NoirTranspiler(z.Circuit.parse(one)).transpile_circuit() == (
        "fn main(s: Field) -> pub Field {\n    let mut o: Field = 0;\n    o = (s * s);\n    o\n}\n")
    none = '(circuit (vars (var 0 "v" field local (init 0))) (instrs (decl 0 (const "1"))))'
    assert z.NoirTranspiler(z.Circuit.parse(none)).transpile_circuit() == "fn main() {\n    let mut v: Field = 1;\n}\n"


@pytest.mark.parametrize("expr,what", [
    ('(lt (var 0) (const "1"))', "<"), ('(band (var 0) (const "1"))', "&"), ('(bnot (var 0))', "~"),
    ('(pow (var 0) (const "3"))', "**"), ('(idiv (var 0) (const "3"))', "\\"), ('(shl (var 0) (const "3"))', "<<")])
def test_operators_without_a_field_counterpart_are_refused(expr, what):
    txt = '(circuit (vars (var 0 "v" field local (init 0))) (instrs (decl 0 (const "1")) (assign 0 %s)))' % expr
    with pytest.raises(z.ZkirError) as e:
        z.NoirTranspiler(z.Circuit.parse(txt)).transpile_circuit()
    assert e.value.code == api.E_UNSUPPORTED and what in str(e.value)


def test_loops_noir_cannot_express_are_refused():
    loop_var_written = '''(circuit (vars (var 0 "i" field local (init 0))) (instrs
        (for (decl 0 (const "0")) (le (var 0) (const "2")) (assign 0 (add (var 0) (const "1"))) (body (assign 0 (const "5"))))))'''
    while_loop = '''(circuit (vars (var 0 "w" field local (init 0))) (instrs (decl 0 (const "0"))
        (while (eq (var 0) (const "0")) (body (assign 0 (const "1"))))))'''
    for txt in (loop_var_written, while_loop):
        with pytest.raises(z.ZkirError) as e:
            z.NoirTranspiler(z.Circuit.parse(txt)).transpile_circuit()
        assert e.value.code == api.E_UNSUPPORTED


def test_arithmetic_circuits_of_the_extended_generator_all_transpile():
    g = z.default_generator_config()
    g.min_instructions = g.max_instructions = 60
    g.forced_inputs, g.inputs_readable, g.arithmetic_only, g.quadratic_aware = 4, 1, 1, 1
    g.no_loops, g.safe_array_index = 1, 1
    for seed in range(1, 40):
        c = z.Generator(seed, g).generate()
        src = z.NoirTranspiler(c).transpile_circuit()
        assert src.startswith("fn main(") and src.endswith("}\n")
        # one assignment line per constraint of the Circom text
        assert src.count(" = ") >= z.CircomTranspiler(c).transpile_circuit().count("<==")
