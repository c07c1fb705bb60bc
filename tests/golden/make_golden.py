#!/usr/bin/env python3
"""Writes the fixtures of tests/golden/.  Run from the repository root:
    python tests/golden/make_golden.py

The reference (razn-v/zkir) has no tests, golden vectors or fixtures of its own, and
cannot be built here (Rust; no cargo), so these vectors come from independent
definitions:
  fr_kat.json    Python big-integer arithmetic (oracle/fr_ref.py) on chosen operands
  rng_kat.json   the xorshift of src/rng.rs:18-24 restated in oracle/zkir_ref.py; the
                 first five values of seeds 0x1, 0x2a and 0x9e3779b97f4a7c15 are also
                 written out by hand in tests/test_oracle.py
  circuits.json  circuits drawn by the Python restatement of src/generator.rs and their
                 Circom text by the restatement of src/circom.rs
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fr_ref as F  # noqa: E402
from oracle import zkir_ref as R  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def fr_kat():
    rnd = random.Random(20240601)
    edge = [0, 1, 2, 3, F.P - 1, F.P - 2, F.HALF, F.HALF + 1, F.HALF - 1, (1 << 64) - 1, 1 << 64, 1 << 253,
            ((1 << 254) - 1) % F.P, (1 << 128) - 1, 31, 32, 33, 63, 64, 252, 253, 254, 255, 256, F.P - 1 - 252,
            F.P - 253, F.P - 254, F.P - 255, F.R, F.R2]
    out = {}
    for code, fn in sorted(list(F.BINARY.items()) + list(F.UNARY.items())):
        rows = []
        pairs = [(a, b) for a in edge[:12] for b in edge[:6]]
        pairs += [(rnd.choice(edge), rnd.choice(edge)) for _ in range(40)]
        pairs += [(rnd.randrange(F.P), rnd.choice(edge)) for _ in range(30)]
        pairs += [(rnd.randrange(F.P), rnd.randrange(F.P)) for _ in range(40)]
        pairs += [(rnd.randrange(F.P), rnd.randrange(300)) for _ in range(20)]
        for a, b in pairs:
            try:
                r, fault = (fn(a, b) if code in F.BINARY else fn(a)), 0
            except F.Fault:
                r, fault = 0, 1
            rows.append([hex(a), hex(b), hex(r), fault])
        out[F.OP_NAMES[code]] = {"code": code, "rows": rows}
    return out


def rng_kat():
    out = {}
    for seed in [1, 0x2A, 0x9E3779B97F4A7C15, 12345, (1 << 64) - 1, 1 << 63]:
        r = R.Rng(seed)
        nxt = [hex(r.next()) for _ in range(16)]
        r = R.Rng(seed)
        rand = [r.rand(lo, hi) for lo, hi in [(0, 1), (1, 10), (15, 30), (7, 7), (0, 51), (0, 60)] * 3]
        r = R.Rng(seed)
        out[hex(seed)] = {"next": nxt, "rand": rand, "rand_string8": r.rand_string(8)}
    return out


def circuits():
    out = {"reference": {}, "extended": {}}
    for seed in [1, 2, 3, 42, 12345, 0xDEADBEEF, 0x9E3779B97F4A7C15]:
        c = R.Generator(R.Rng(seed)).generate()
        out["reference"][hex(seed)] = {"text": R.print_text(c), "circom": R.transpile_circuit(c)}
    ext = dict(min_instructions=60, max_instructions=60, forced_inputs=4, inputs_readable=1, arithmetic_only=1,
               quadratic_aware=1, no_loops=1, safe_array_index=1)
    for seed in [7, 77, 777]:
        c = R.Generator(R.Rng(seed), **ext).generate()
        out["extended"][hex(seed)] = {"config": ext, "text": R.print_text(c), "circom": R.transpile_circuit(c)}
    return out


if __name__ == "__main__":
    for name, fn in [("fr_kat.json", fr_kat), ("rng_kat.json", rng_kat), ("circuits.json", circuits)]:
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(fn(), f, indent=0, sort_keys=True)
        print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes")
