"""The cross-check bundles under tests/golden/crosscheck (made on an MI355X by make_bundles.py there: the expected
values are the GPU's) against the big-integer interpretation of the same circuits: the Circom text is what the
Python restatement of src/circom.rs prints, every expected output is what the interpreter computes for the bundle's
inputs, every refused circuit is refused with the same class.  What the bundles cannot be checked against here is
circom itself: that is what they are for."""
import glob
import json
import os

from oracle import zkir_ref as R

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "golden", "crosscheck")


def _make_bundles():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_bundles", os.path.join(ROOT, "make_bundles.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _circuit_of(name, mb):
    kind, _, rest = name.partition("_")
    if kind == "reference":
        return R.Generator(R.Rng(int(rest[4:]))).generate(), {}
    if kind == "extended":
        seed = int(rest[4:])
        cfg = dict(min_instructions=60, max_instructions=60, forced_inputs=4, inputs_readable=1, arithmetic_only=1,
                   quadratic_aware=1, no_loops=seed % 2, safe_array_index=1)
        return R.Generator(R.Rng(seed), **cfg).generate(), dict(max_iterations=300)
    if kind == "deep":
        cfg = dict(min_instructions=30, max_instructions=40, max_expr_depth=3, inputs_readable=1, safe_array_index=1, no_loops=1,
                   quadratic_aware=1, forced_inputs=3)
        return R.Generator(R.Rng(int(rest[4:])), **cfg).generate(), {}
    assert kind == "operator"
    return R.parse_text(mb.operator_circuit(rest)), {}


def _values(circ, doc, role):
    """flatten {name: value or list} to the order the interpreter keeps inputs / outputs in"""
    out = []
    for v in circ["vars"]:
        if v["role"] == role:
            x = doc[v["name"]]
            out += [int(t) for t in x] if isinstance(x, list) else [int(x)]
    return out


def test_bundles_agree_with_the_big_integer_interpreter():
    mb = _make_bundles()
    with open(os.path.join(ROOT, "index.json")) as f:
        index = json.load(f)
    assert len(index) >= 20
    n_vectors = n_refused = 0
    for name, status in sorted(index.items()):
        d = os.path.join(ROOT, name)
        circ, opts = _circuit_of(name, mb)
        with open(os.path.join(d, "circuit.circom")) as f:
            assert f.read() == R.transpile_circuit(circ), name
        with open(os.path.join(d, "verdict.json")) as f:
            verdict = json.load(f)
        assert os.path.exists(os.path.join(d, "run.sh"))
        if not verdict["accepted"]:
            n_refused += 1
            n_in = sum((v["array_len"] or 1) for v in circ["vars"] if v["role"] == "input")
            try:
                R.Interpreter(circ, [0] * n_in, max_iterations=opts.get("max_iterations", 4096)).run()
                assert False, name + ": the interpreter accepts what lower refused"
            except R.Rejected as e:
                assert e.cls == verdict["class"], (name, e.cls, verdict)
            continue
        for path in sorted(glob.glob(os.path.join(d, "input_*.json"))):
            k = path[:-5].rsplit("_", 1)[1]
            with open(path) as f:
                inputs = _values(circ, json.load(f), "input")
            for suffix, variant in (("", 0), (".bnot256", 1)):
                exp = os.path.join(d, "expected_%s%s.json" % (k, suffix))
                if not os.path.exists(exp):
                    assert suffix, exp
                    continue
                it = R.Interpreter(circ, inputs, bnot_variant=variant, **opts).run()
                with open(exp) as f:
                    assert it.outputs == _values(circ, json.load(f), "output"), (name, k, suffix)
                assert it.unsat() == []
                n_vectors += 1
    assert n_vectors >= 60 and n_refused >= 5
    # the two readings of `~` differ on these constants: the bundle can tell them apart
    d = os.path.join(ROOT, "operator_bnot")
    with open(os.path.join(d, "expected_0.json")) as f, open(os.path.join(d, "expected_0.bnot256.json")) as g:
        assert json.load(f) != json.load(g)
