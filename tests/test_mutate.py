"""Tape mutations stay loadable and (almost always) break a constraint, per the big-integer oracle."""
import random

import zkir_amd as z
from oracle import fr_ref as F
from zkir_amd import mutate


def test_mutations_break_constraints_and_stay_valid():
    g = z.default_generator_config()
    g.min_instructions = g.max_instructions = 120
    g.forced_inputs, g.inputs_readable, g.arithmetic_only, g.quadratic_aware = 4, 1, 1, 1
    g.no_loops, g.safe_array_index = 1, 1
    rnd = random.Random(3)
    broken = tried = 0
    for seed in range(1, 25):
        tape = z.Lowerer(z.Generator(seed, g).generate()).lower()
        vec = [rnd.randrange(F.P) for _ in range(tape.hdr.n_inputs)]
        assert F.eval_tape_one(tape.ops(), tape.consts(), vec)[1] == []
        for f in mutate.MUTATIONS:
            t, what = f(tape, rnd)
            if t is None:
                continue
            tried += 1
            assert what and bytes(t.hdr) == bytes(tape.hdr) and len(t.blob()) == len(tape.blob())
            broken += bool(F.eval_tape_one(t.ops(), t.consts(), vec)[1])
        t, what = mutate.mutate(tape, rnd)
        assert t is not None
    assert tried >= 40 and broken >= tried * 0.6  # a mutation can land on a value no constraint uses, or pick an equal constant
