"""Tape mutations for self-tests of a campaign: break a lowered circuit in a small, known way and
expect the evaluator to report it.  (The reference names mutation support as a goal, README.md:12,
and has no code for it; these mutations act on the tape, after lowering.)

Every mutation keeps the tape valid (indices stay inside the header's counts), so a mutated tape
loads like any other; whether it still satisfies its constraints is what the evaluator finds out.
"""
import struct

from . import api

OP_ADD, OP_SUB, OP_MUL, OP_CHECK = 16, 17, 18, 48
KIND_SHIFT, K_CONST = 29, 2


def _rebuild(tape, ops):
    blob = tape.blob()
    return api.Tape.from_parts(tape.hdr, b"".join(struct.pack("<4I", *op) for op in ops) + blob[16 * len(ops):])


def swap_opcode(tape, rnd, pairs=((OP_MUL, OP_ADD), (OP_ADD, OP_SUB), (OP_SUB, OP_ADD))):
    """Replace one + - * by another.  Returns (mutated tape, description) or (None, None) when the
    tape has no such op."""
    ops = tape.ops()
    sites = [(i, new) for i, op in enumerate(ops) for old, new in pairs if (op[0] & 0xFF) == old]
    if not sites:
        return None, None
    i, new = rnd.choice(sites)
    old = ops[i][0] & 0xFF
    ops[i] = (new,) + ops[i][1:]
    return _rebuild(tape, ops), "op %d: opcode %d -> %d" % (i, old, new)


def swap_check_operand(tape, rnd):
    """Make one CHECK compare against another constant of the pool (needs two constants)."""
    ops = tape.ops()
    if tape.hdr.n_consts < 2:
        return None, None
    sites = [i for i, op in enumerate(ops) if (op[0] & 0xFF) == OP_CHECK]
    if not sites:
        return None, None
    i = rnd.choice(sites)
    code, c, a, b = ops[i]
    k = rnd.randrange(tape.hdr.n_consts)
    ops[i] = (code, (K_CONST << KIND_SHIFT) | k, a, b)
    return _rebuild(tape, ops), "op %d: check against constant %d" % (i, k)


MUTATIONS = (swap_opcode, swap_check_operand)


def mutate(tape, rnd):
    """One random mutation that applies to this tape: (tape, description), or (None, None)."""
    for f in rnd.sample(MUTATIONS, len(MUTATIONS)):
        t, what = f(tape, rnd)
        if t is not None:
            return t, what
    return None, None
