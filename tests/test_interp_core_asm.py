"""The generated interpreter stream (tools/gen_interp_asm.py), executed instruction by
instruction on the model in isa_model.py for single lanes, against the big-integer tape
interpreter of the oracle.  Covers the arithmetic, the register plan, the waits and, above
all, that every memory access stays inside its buffer."""
import importlib.util
import os
import random

import pytest

from oracle import fr_ref as F
from tests import isa_model as IM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))



def one_word_of(consts):
    """Operand word of the constant 1 in a tape's pool, as the runtime finds it at upload (0xffffffff: none)."""
    for i, c in enumerate(consts):
        if c == 1:
            return (2 << 29) | i
    return 0xFFFFFFFF

def _gen():
    spec = importlib.util.spec_from_file_location("gen_interp_asm", os.path.join(ROOT, "tools", "gen_interp_asm.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


G = _gen()
PROG, LABELS = IM.parse_program(G.build())
R = (1 << 256) % F.P

OPS_BASE, CST_BASE, GS_BASE = 0x7F0000001000, 0x7F0010000000, 0x7F1000000000
IN_BASE, OUT_BASE, LDS_BASE = 0x7E0000000400, 0x7D0000000800, 16


def run_lane(ops, consts, inputs, n_outputs, n_lslots, n_gslots, vec, wave=0, lane=0, live=True, want_outputs=True,
             block=3, victim=True, n_xslots=0):
    """inputs: list over vectors of lists of ints.  Returns (outputs list or None, unsat, first, Lane)."""
    n_vec = len(inputs)
    n_inputs = len(inputs[0]) if inputs else 0
    mem = IM.Memory()
    words = []
    for code, dst, a, b in ops:
        words += [code, dst, a, b]
    mem.map("ops", OPS_BASE, max(16, 16 * len(ops)), words=words)
    cw = []
    for c in consts:
        cw += F.limbs32(c * R % F.P)
    mem.map("consts", CST_BASE, max(32, 32 * len(consts)), words=cw)
    gs_bytes = n_gslots * 8192
    gs = GS_BASE + block * gs_bytes
    mem.map("gslots", gs, max(gs_bytes, 16), writable=True)
    iw = [0] * (n_inputs * 8 * n_vec)
    for t, vecvals in enumerate(inputs):
        for s, x in enumerate(vecvals):
            for l, w in enumerate(F.limbs32(x)):
                iw[(s * 8 + l) * n_vec + t] = w
    mem.map("inputs", IN_BASE, max(16, 4 * len(iw)), words=iw)
    out_region = mem.map("outputs", OUT_BASE, max(16, 32 * n_outputs * n_vec), writable=True)
    v = vec if live else 0
    bit = 1 << lane
    operands = {
        "%[ops]": OPS_BASE, "%[n_ops]": len(ops), "%[consts]": CST_BASE, "%[n_consts]": len(consts), "%[one_word]": one_word_of(consts),
        "%[rot]": 0xFFFFFFFF, "%[rot_shift]": 12,
        "%[gslots]": gs, "%[gslot_bytes]": gs_bytes,
        "%[lds_base]": LDS_BASE + (8192 if victim else 0),  # the victim row, if any, comes first
        "%[evict_below]": 0x20000000 if victim else 0,
        "%[in_base]": IN_BASE, "%[out_base]": OUT_BASE if want_outputs else 0,
        "%[n4]": 4 * n_vec, "%[live]": (0x8000000000000001 | bit) if live else (0x8000000000000001 & ~bit),
        "%[v_lo]": wave * 2048 + lane * 16, "%[v_vec4]": 4 * v,
        "%[xbase]": LDS_BASE + (n_lslots + (1 if victim else 0)) * 8192, "%[v_x]": lane * 16,
    }
    ln = IM.Lane(mem, LDS_BASE + (n_lslots + (1 if victim else 0)) * 8192 + n_xslots * 2048, lane, operands, set(range(G.VB, G.V_LAST + 1)),
                 set(range(G.S_FIRST, G.S_LAST + 1)), result_operands=("%[unsat]", "%[first]"))
    ln.run(PROG, LABELS)
    outs = None
    if want_outputs and live:
        outs = []
        for s in range(n_outputs):
            limbs = [out_region.words.get(OUT_BASE + 4 * ((s * 8 + l) * n_vec + v)) for l in range(8)]
            outs.append(None if None in limbs else F.from_limbs32(limbs))
    elif not live:
        assert not out_region.words, "an idle lane stored outputs"
    return outs, ln.operand_written["%[unsat]"], ln.operand_written["%[first]"], ln


def oracle(ops, consts, vecvals, n_outputs):
    outs, unsat, faults = F.eval_tape_one(ops, consts, vecvals)
    assert faults == 0
    return [outs.get(s) for s in range(n_outputs)], unsat


def check_tape(ops, consts, inputs, n_outputs, n_lslots, n_gslots, picks):
    for vec, wave, lane, live in picks:
        want_out, want_unsat = oracle(ops, consts, inputs[vec if live else 0], n_outputs)
        got_out, unsat, first, _ = run_lane(ops, consts, inputs, n_outputs, n_lslots, n_gslots, vec, wave, lane, live)
        if live:
            assert got_out == want_out
            assert unsat == len(want_unsat)
            assert first == (want_unsat[0] if want_unsat else 0xFFFFFFFF)
        else:
            assert unsat == 0 and first == 0xFFFFFFFF


def O(kind, idx=0):
    return F.operand(kind, idx)


def test_every_operand_kind_and_op():
    rnd = random.Random(11)
    consts = [0, 1, F.P - 1, rnd.randrange(F.P), (1 << 253) + 12345]
    NONE, ACC = O(F.K_NONE), O(F.K_ACC)
    ops = [
        (F.OP_INPUT, O(F.K_LSLOT, 0), O(F.K_INPUT, 0), NONE),
        (F.OP_INPUT, O(F.K_GSLOT, 0), O(F.K_INPUT, 1), NONE),
        (F.OP_INPUT, NONE, O(F.K_INPUT, 2), NONE),
        (F.OP_MUL, O(F.K_LSLOT, 1), ACC, O(F.K_LSLOT, 0)),             # acc * lds
        (F.OP_MUL, O(F.K_GSLOT, 1), O(F.K_GSLOT, 0), O(F.K_CONST, 3)),  # device * const
        (F.OP_ADD, O(F.K_GSLOT, 2), ACC, ACC),                          # acc + acc
        (F.OP_ADD, NONE, O(F.K_LSLOT, 1), O(F.K_GSLOT, 1)),
        (F.OP_SUB, O(F.K_LSLOT, 2), ACC, O(F.K_CONST, 2)),              # acc - (p-1)
        (F.OP_SUB, NONE, O(F.K_CONST, 0), ACC),                         # 0 - acc
        (F.OP_SUB, O(F.K_LSLOT, 3), O(F.K_GSLOT, 2), O(F.K_LSLOT, 2)),
        (F.OP_NEG, O(F.K_GSLOT, 3), ACC, NONE),
        (F.OP_NEG, NONE, O(F.K_CONST, 0), NONE),                        # -0
        (F.OP_COPY, O(F.K_LSLOT, 0), O(F.K_GSLOT, 3), NONE),
        (F.OP_COPY, O(F.K_GSLOT, 0), ACC, NONE),
        (F.OP_COPY, NONE, O(F.K_CONST, 4), NONE),
        (F.OP_MUL, NONE, ACC, ACC),                                     # square of a constant
        (F.OP_OUTPUT, 0, ACC, NONE),
        (F.OP_OUTPUT, 2, O(F.K_LSLOT, 3), NONE),
        (F.OP_OUTPUT, 1, O(F.K_GSLOT, 1), NONE),
        (F.OP_OUTPUT, 3, O(F.K_CONST, 3), NONE),
        (F.OP_NOP, 0, 0, 0),
        (F.OP_MUL, O(F.K_LSLOT, 1), O(F.K_LSLOT, 0), O(F.K_GSLOT, 0)),
        (F.OP_CHECK, ACC, O(F.K_LSLOT, 0), O(F.K_GSLOT, 0)),            # holds
        (F.OP_CHECK, O(F.K_LSLOT, 1), ACC, O(F.K_CONST, 1)),            # acc*1 == lds1: holds
        (F.OP_CHECK, O(F.K_CONST, 3), O(F.K_LSLOT, 0), O(F.K_LSLOT, 0)),  # fails (almost surely)
        (F.OP_CHECK, O(F.K_GSLOT, 1), O(F.K_GSLOT, 0), O(F.K_CONST, 3)),
        (F.OP_CHECK, O(F.K_CONST, 0), O(F.K_CONST, 0), ACC),            # 0*acc == 0: holds
        # scaled values: the input times 2^-256 in, times a plain value, plus a scaled one, times 2^256 out
        (F.OP_INPUT_SCALED, O(F.K_GSLOT, 2), O(F.K_INPUT, 1), NONE),
        (F.OP_INPUT_SCALED, NONE, O(F.K_INPUT, 0), NONE),
        (F.OP_MUL, O(F.K_LSLOT, 2), ACC, O(F.K_GSLOT, 0)),
        (F.OP_ADD, NONE, ACC, O(F.K_GSLOT, 2)),
        (F.OP_OUTPUT_SCALED, 4, ACC, NONE),
        (F.OP_OUTPUT_SCALED, 5, O(F.K_GSLOT, 2), NONE),
        (F.OP_OUTPUT_SCALED, 6, O(F.K_LSLOT, 2), NONE),
        (F.OP_OUTPUT_SCALED, 7, O(F.K_CONST, 4), NONE),
        (F.OP_SUB, NONE, ACC, O(F.K_GSLOT, 2)),
        (F.OP_CHECK, ACC, O(F.K_LSLOT, 2), O(F.K_CONST, 1)),            # (scaled product) * 1 == it: holds
    ]
    inputs = [[rnd.randrange(F.P) for _ in range(3)] for _ in range(70)]
    inputs[0] = [0, F.P - 1, 1]
    inputs[1] = [F.P - 1, F.P - 1, F.P - 1]
    check_tape(ops, consts, inputs, 8, 4, 4, [(0, 0, 0, True), (1, 0, 1, True), (69, 1, 5, True), (33, 3, 63, True),
                                              (5, 2, 7, False)])


def test_outputs_not_requested_and_empty_tape():
    ops = [(F.OP_INPUT, O(F.K_NONE), O(F.K_INPUT, 0), O(F.K_NONE)), (F.OP_OUTPUT, 0, O(F.K_ACC), O(F.K_NONE)),
           (F.OP_CHECK, O(F.K_ACC), O(F.K_ACC), O(F.K_ACC))]
    inputs = [[5], [1], [0]]
    outs, unsat, first, _ = run_lane(ops, [], inputs, 1, 0, 0, 0, want_outputs=False)
    assert outs is None and unsat == 1 and first == 0  # 5*5 != 5
    outs, unsat, first, _ = run_lane(ops, [], inputs, 1, 0, 0, 1, lane=1)
    assert outs == [1] and unsat == 0 and first == 0xFFFFFFFF
    outs, unsat, first, ln = run_lane([], [], inputs, 0, 0, 0, 0)
    assert unsat == 0 and first == 0xFFFFFFFF and ln.steps < 80


def test_strictness_of_the_model():
    # an op that names a device slot beyond the region must trip the model, not pass silently
    ops = [(F.OP_INPUT, O(F.K_GSLOT, 2), O(F.K_INPUT, 0), O(F.K_NONE))]
    with pytest.raises(IM.ModelError):
        run_lane(ops, [], [[3]], 0, 0, 2, 0)
    ops = [(F.OP_COPY, O(F.K_NONE), O(F.K_LSLOT, 1), O(F.K_NONE))]
    with pytest.raises(IM.ModelError):
        run_lane(ops, [], [[3]], 0, 1, 0, 0)


def _lowered(seed, statements, **opts):
    from zkir_amd import api
    g = api.default_generator_config()
    g.min_instructions = g.max_instructions = statements
    g.forced_inputs, g.inputs_readable, g.arithmetic_only, g.quadratic_aware = 4, 1, 1, 1
    g.no_loops, g.safe_array_index = 1, 1
    lo = api.default_lower_options()
    for k, v in opts.items():
        setattr(lo, k, v)
    return api.Lowerer(api.Generator(seed, g).generate(), lo).lower()


@pytest.mark.parametrize("seed,statements,opts", [
    (101, 60, {}),
    (102, 120, {"n_lslots": 2}),
    (103, 120, {"n_lslots": 0}),
    (104, 90, {"use_acc": 0}),
    (105, 150, {"n_lslots": 8, "window_lifetime": 8}),
])
def test_generated_circuits(seed, statements, opts):
    tape = _lowered(seed, statements, **opts)
    h = tape.hdr
    ops, consts = tape.ops(), tape.consts()
    assert all((c & 0xFF) in (0, 1, 2, 3, 4, 8, 9, 16, 17, 18, 48) for c, _, _, _ in ops)
    rnd = random.Random(seed)
    n_vec = 300
    inputs = [[rnd.randrange(F.P) for _ in range(h.n_inputs)] for _ in range(n_vec)]
    picks = [(0, 0, 0, True), (299, 0, 43, True), (rnd.randrange(n_vec), 2, 17, True), (77, 3, 63, False)]
    check_tape(ops, consts, inputs, h.n_outputs, h.n_lslots, h.n_gslots, picks)


def test_a_broken_witness_is_counted():
    tape = _lowered(106, 80)
    h = tape.hdr
    ops, consts = tape.ops(), tape.consts()
    # turn the first product into a sum: the constraints that use it stop holding
    k = next(i for i, o in enumerate(ops) if (o[0] & 0xFF) == F.OP_MUL)
    ops[k] = (F.OP_ADD,) + ops[k][1:]
    rnd = random.Random(7)
    inputs = [[rnd.randrange(F.P) for _ in range(h.n_inputs)] for _ in range(3)]
    want_out, want_unsat = oracle(ops, consts, inputs[2], h.n_outputs)
    assert want_unsat
    got_out, unsat, first, _ = run_lane(ops, consts, inputs, h.n_outputs, h.n_lslots, h.n_gslots, 2, 1, 9)
    assert got_out == want_out and unsat == len(want_unsat) and first == want_unsat[0]


def test_check_against_the_constant_one_multiplies_nothing():
    """`value * 1 == signal`: with the pool's 1 as right factor the left factor is compared as it is, from every
    operand kind, holding and failing alike; the same checks against another constant run the product."""
    rnd = random.Random(5)
    NONE, ACC = O(F.K_NONE), O(F.K_ACC)

    def tape(k):  # k: index of the right factor in the pool
        return [
            (F.OP_INPUT, O(F.K_LSLOT, 0), O(F.K_INPUT, 0), NONE),
            (F.OP_INPUT, O(F.K_GSLOT, 0), O(F.K_INPUT, 1), NONE),
            (F.OP_CHECK, O(F.K_GSLOT, 0), O(F.K_GSLOT, 0), O(F.K_CONST, k)),  # device slot against itself: holds
            (F.OP_CHECK, O(F.K_LSLOT, 0), O(F.K_GSLOT, 0), O(F.K_CONST, k)),  # against the other input: fails
            (F.OP_CHECK, O(F.K_LSLOT, 0), O(F.K_LSLOT, 0), O(F.K_CONST, k)),  # LDS slot: holds
            (F.OP_COPY, NONE, O(F.K_LSLOT, 0), NONE),
            (F.OP_CHECK, ACC, ACC, O(F.K_CONST, k)),                          # accumulator against itself: holds
            (F.OP_CHECK, O(F.K_GSLOT, 0), ACC, O(F.K_CONST, k)),              # accumulator against the other: fails
            (F.OP_CHECK, ACC, O(F.K_LSLOT, 0), O(F.K_CONST, k)),              # holds
            (F.OP_CHECK, O(F.K_CONST, 2), O(F.K_CONST, 2), O(F.K_CONST, k)),  # constant: holds
            (F.OP_MUL, NONE, O(F.K_GSLOT, 0), O(F.K_CONST, k)),               # a product by 1 is still a product
            (F.OP_CHECK, ACC, O(F.K_GSLOT, 0), O(F.K_CONST, k)),              # holds, with the slot still in the a registers
            (F.OP_OUTPUT, 0, O(F.K_GSLOT, 0), NONE),
        ]
    inputs = [[rnd.randrange(F.P), rnd.randrange(F.P)] for _ in range(66)]
    inputs[1] = [F.P - 1, 0]
    picks = [(0, 0, 0, True), (1, 1, 1, True), (65, 2, 40, True), (9, 3, 63, False)]
    consts = [7, 1, F.P - 1]
    check_tape(tape(1), consts, inputs, 1, 1, 1, picks)
    # the same tape with the 1 in a pool that holds it at another place, and in a pool that holds none: x * 7
    check_tape(tape(0), [1, 5, 9], inputs, 1, 1, 1, picks)
    check_tape(tape(0), consts, inputs, 1, 1, 1, picks)
    _, unsat, first, by_one = run_lane(tape(1), consts, inputs, 1, 1, 1, 0)
    assert (unsat, first) == (2, 1)
    _, _, _, by_seven = run_lane(tape(0), consts, inputs, 1, 1, 1, 0)
    assert by_one.steps < by_seven.steps - 7 * 150  # seven products not made


def test_operands_of_the_next_op_are_sent_for_inside_a_product():
    """Inside the product of a MUL, and of a CHECK whose right-hand side is the accumulator, the stream sends for the
    next op's a and b operands; that op finds them by their tags and waits for them.  Not when the next op reads what
    this op is about to store, and never for the last op.  Same results, and no load is issued twice."""
    rnd = random.Random(21)
    NONE, ACC = O(F.K_NONE), O(F.K_ACC)
    G = lambda i: O(F.K_GSLOT, i)
    head = [(F.OP_INPUT, G(i), O(F.K_INPUT, i), NONE) for i in range(5)]
    ops = head + [
        (F.OP_MUL, G(5), G(0), G(1)),      # sends for G2, G3 (the next op's operands)
        (F.OP_MUL, G(6), G(2), G(3)),      # next op reads G6 = this op's destination: nothing sent for
        (F.OP_ADD, NONE, G(6), G(4)),
        (F.OP_OUTPUT, 0, ACC, NONE),
        (F.OP_MUL, NONE, G(0), G(4)),      # a is not in the registers any more, b is; sends for G5 and G1
        (F.OP_CHECK, ACC, G(1), G(5)),     # wrong on purpose; right-hand side in the accumulator: sends for G2 (b) and G6 (a)
        (F.OP_SUB, NONE, G(6), G(2)),
        (F.OP_OUTPUT, 1, ACC, NONE),
        (F.OP_MUL, G(7), ACC, G(3)),       # the next op is unary: nothing sent for
        (F.OP_OUTPUT, 2, G(7), NONE),
        (F.OP_MUL, NONE, G(7), G(7)),      # the last op
    ]
    inputs = [[rnd.randrange(F.P) for _ in range(5)] for _ in range(3)]
    check_tape(ops, [], inputs, 3, 0, 8, [(0, 0, 0, True), (2, 3, 63, True), (1, 1, 7, False)])
    _, unsat, first, ln = run_lane(ops, [], inputs, 3, 0, 8, 1, victim=False)
    assert unsat == 1 and first == 0
    # values loaded from device slots: G0 G1 | G2 G3 | G6 G4 | G0 | G5 G1 | G2 G6 | G3 | G7 (OUTPUT, into the a
    # registers) | G7 as b of the last MUL (its a is found in the a registers): no value is loaded twice in a row
    assert ln.counts["buffer_load_dwordx4"] == 2 * 14


def test_inc_file_is_current():
    path = os.path.join(ROOT, "zkir_amd", "csrc", "hip", "interp_core_asm.inc")
    assert open(path).read() == G.render(G.build()), "run tools/gen_interp_asm.py"


def test_register_discipline():
    """Static checks over the text: no operand of the statement is ever a destination except
    the two results in the last instructions; every named register is in the clobber list."""
    import re
    lines = [l for l in G.build() if not l.endswith(":")]
    for i, l in enumerate(lines):
        op, _, rest = l.partition(" ")
        args = [x.strip() for x in rest.split(",")] if rest else []
        writes_first = not (op.startswith("s_cmp") or op.startswith("s_cbranch") or op in ("s_branch", "s_nop", "s_waitcnt")
                            or op.startswith("buffer_store") or op.startswith("ds_write"))
        if writes_first and args and args[0].startswith("%["):
            assert i >= len(lines) - 2 and args[0] in ("%[unsat]", "%[first]"), l
        if op.startswith("v_") and len(args) > 1 and args[1].startswith("%[") and "_co_" in op:
            raise AssertionError("carry destination is an operand: " + l)
        for m in re.finditer(r"\bv(\d+)\b", l):
            assert G.VB <= int(m.group(1)) <= G.V_LAST, l
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]", l):
            assert G.VB <= int(m.group(1)) and int(m.group(2)) <= G.V_LAST, l
        for m in re.finditer(r"\bs(\d+)\b", l):
            assert G.S_FIRST <= int(m.group(1)) <= G.S_LAST, l
        for m in re.finditer(r"\bs\[(\d+):(\d+)\]", l):
            assert G.S_FIRST <= int(m.group(1)) and int(m.group(2)) <= G.S_LAST, l


def test_reduction_zeroes_its_addend_halves_itself():
    """The high halves of the reduction stream's 64-bit addends must be zero when its multiply-adds read them, and the
    product stream uses the same registers as scratch: every copy of the reduction starts by zeroing the eight of them,
    and writes none of them afterwards."""
    import re
    zeros = {int(G.Z(j)[1:]) for j in range(8)}
    a = G.Asm()
    G.redc_stream(a, G.A8)
    lines = a.lines
    assert lines[:8] == ["v_mov_b32 %s, 0" % G.Z(j) for j in range(8)]
    for raw in lines[8:]:
        op, _, rest = raw.partition(" ")
        if op.startswith("s_"):
            continue
        dst = rest.split(",")[0].strip()
        regs = set()
        m = re.fullmatch(r"v(\d+)", dst)
        if m:
            regs = {int(m.group(1))}
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", dst)
        if m:
            regs = set(range(int(m.group(1)), int(m.group(2)) + 1))
        if regs & zeros:  # a multiply-add writes the pair {E, Z}: its high half is rewritten by the adds that follow...
            assert op == "v_mad_u64_u32", raw
    # ...so what matters is the value each multiply-add reads there: the model run below would catch a wrong one
    rnd = random.Random(5)
    x = rnd.randrange(F.P)
    ops = [(F.OP_INPUT, O(F.K_NONE), O(F.K_INPUT, 0), O(F.K_NONE)), (F.OP_MUL, O(F.K_NONE), O(F.K_ACC), O(F.K_ACC)),
           (F.OP_OUTPUT, 0, O(F.K_ACC), O(F.K_NONE)), (F.OP_OUTPUT, 1, O(F.K_ACC), O(F.K_NONE))]
    outs, unsat, first, _ = run_lane(ops, [], [[x]], 2, 0, 0, 0)
    assert outs == [x * x % F.P] * 2


def test_operand_registers_are_reused_and_invalidated():
    """An operand whose value already sits in the a or b registers is not loaded again; a store
    to its slot, or anything else that overwrites those registers, forces the reload."""
    rnd = random.Random(12)
    NONE, ACC = O(F.K_NONE), O(F.K_ACC)
    G0, G1, G2, L0 = O(F.K_GSLOT, 0), O(F.K_GSLOT, 1), O(F.K_GSLOT, 2), O(F.K_LSLOT, 0)
    head = [(F.OP_INPUT, G0, O(F.K_INPUT, 0), NONE), (F.OP_INPUT, G1, O(F.K_INPUT, 1), NONE),
            (F.OP_INPUT, G2, O(F.K_INPUT, 2), NONE)]
    reuse = head + [
        (F.OP_MUL, L0, G0, G1),          # loads G0, G1; inside the product G2 is sent for (the next op's b)
        (F.OP_ADD, NONE, ACC, G2),       # finds G2 in the b registers
        (F.OP_OUTPUT, 0, ACC, NONE),
        (F.OP_SUB, NONE, ACC, G2),       # b is still there
        (F.OP_CHECK, ACC, G0, G1),       # a is still there, b is reloaded
        (F.OP_CHECK, L0, G0, G1),        # a and b still there (the product went to scratch registers); c (L0) replaces b
        (F.OP_MUL, NONE, O(F.K_CONST, 0), L0),  # L0 sits in the b registers
        (F.OP_OUTPUT, 1, ACC, NONE),
    ]
    inputs = [[rnd.randrange(F.P) for _ in range(3)] for _ in range(2)]
    consts = [rnd.randrange(F.P)]
    want_out, want_unsat = oracle(reuse, consts, inputs[1], 2)
    got_out, unsat, first, ln = run_lane(reuse, consts, inputs, 2, 1, 3, 1, lane=3)
    assert got_out == want_out and unsat == len(want_unsat) == 0
    # device-slot loads: G0, G1, G2 = 3 values = 6 instructions;
    # G1 for the first CHECK comes back from the victim row, where G2's arrival had pushed it
    assert ln.counts["buffer_load_dwordx4"] == 6
    assert ln.counts["ds_read_b128"] == 4  # G1 from the victim row, L0 as c of the second CHECK
    # without a victim row (launches with a large window) G1 is loaded again, and LDS holds the window only
    got_out, unsat, first, ln = run_lane(reuse, consts, inputs, 2, 1, 3, 1, lane=3, victim=False)
    assert got_out == want_out and unsat == 0
    assert ln.counts["buffer_load_dwordx4"] == 8 and ln.counts["ds_read_b128"] == 2 and "ds_write_b128" in ln.counts

    stale = head + [
        (F.OP_MUL, NONE, G0, G1),
        (F.OP_ADD, G0, ACC, G2),         # overwrites G0 while its old value sits in the a registers
        (F.OP_MUL, G1, G0, G1),          # must see the new G0; then overwrites G1 (in the b registers)
        (F.OP_SUB, L0, G0, G1),          # must see the new G1
        (F.OP_OUTPUT, 0, ACC, NONE),
        (F.OP_ADD, L0, L0, L0),          # reads L0 twice, then overwrites it
        (F.OP_SUB, NONE, L0, G2),        # must see the new L0
        (F.OP_OUTPUT, 1, ACC, NONE),
        (F.OP_NEG, NONE, G2, NONE),      # zeroes the a registers behind the tag's back
        (F.OP_ADD, NONE, L0, G2),        # a must be reloaded
        (F.OP_OUTPUT, 2, ACC, NONE),
    ]
    want_out, _ = oracle(stale, [], inputs[0], 3)
    for victim in (True, False):
        got_out, _, _, _ = run_lane(stale, [], inputs, 3, 1, 3, 0, wave=1, lane=40, victim=victim)
        assert got_out == want_out


# ---- strand tapes: four wavefronts, one lane of each, sharing LDS and the output buffer -------------
def run_strands(ops, strand_ops, consts, inputs, n_outputs, n_lslots, n_gslots, n_xslots, vec, lane=0, rot=0xFFFFFFFF,
                clock_step=0, prio_logs=None):
    """The same lane of the wavefronts of a workgroup (four or eight: len(strand_ops)), each running its strand of
    `ops` from barrier to barrier in turn, with the operands the kernel around the stream gives them (eight
    wavefronts keep two sets of slot rows: wavefronts 4..7 use the second).  rot: the workgroup's phase in the
    priority turns (0xFFFFFFFF: none); the model's clock advances by clock_step per barrier.
    Returns (outputs, unsat summed over the strands, smallest first-unsat)."""
    n_str = len(strand_ops)
    sets = 2 if n_str == 8 else 1
    n_vec, n_inputs = len(inputs), len(inputs[0])
    mem = IM.Memory()
    words = []
    for code, dst, a, b in ops:
        words += [code, dst, a, b]
    mem.map("ops", OPS_BASE, max(16, 16 * len(ops)), words=words)
    cw = []
    for c in consts:
        cw += F.limbs32(c * R % F.P)
    mem.map("consts", CST_BASE, max(32, 32 * len(consts)), words=cw)
    gs_bytes = n_gslots * 8192
    mem.map("gslots", GS_BASE, max(sets * gs_bytes, 16), writable=True)
    iw = [0] * (n_inputs * 8 * n_vec)
    for t, vecvals in enumerate(inputs):
        for s, x in enumerate(vecvals):
            for l, w in enumerate(F.limbs32(x)):
                iw[(s * 8 + l) * n_vec + t] = w
    mem.map("inputs", IN_BASE, max(16, 4 * len(iw)), words=iw)
    out_region = mem.map("outputs", OUT_BASE, max(16, 32 * n_outputs * n_vec), writable=True)
    shared_lds = {}
    lanes, pcs, begin = [], [], 0
    for w in range(n_str):
        operands = {
            "%[ops]": OPS_BASE + 16 * begin, "%[n_ops]": strand_ops[w], "%[consts]": CST_BASE, "%[n_consts]": len(consts), "%[one_word]": one_word_of(consts),
            "%[gslots]": GS_BASE + (w >> 2) * gs_bytes, "%[gslot_bytes]": gs_bytes,
            "%[lds_base]": LDS_BASE + (w >> 2) * n_lslots * 8192, "%[evict_below]": 0,
            "%[in_base]": IN_BASE, "%[out_base]": OUT_BASE, "%[n4]": 4 * n_vec, "%[live]": 0x8000000000000001 | (1 << lane),
            "%[v_lo]": (w & 3) * 2048 + lane * 16, "%[v_vec4]": 4 * vec,
            "%[xbase]": LDS_BASE + sets * n_lslots * 8192, "%[v_x]": lane * 16, "%[rot]": rot, "%[rot_shift]": 12,
        }
        ln = IM.Lane(mem, LDS_BASE + sets * n_lslots * 8192 + n_xslots * 2048, lane, operands, set(range(G.VB, G.V_LAST + 1)),
                     set(range(G.S_FIRST, G.S_LAST + 1)), result_operands=("%[unsat]", "%[first]"))
        ln.lds = shared_lds
        lanes.append(ln)
        pcs.append(0)
        begin += strand_ops[w]
    arrived = [0] * n_str
    while any(pc is not None for pc in pcs):
        for w in range(n_str):
            if pcs[w] is not None:
                lanes[w].clock = (1 << 40) + clock_step * arrived[w]
                pcs[w] = lanes[w].run(PROG, LABELS, pc=pcs[w], stop_at_barrier=True)
                arrived[w] += pcs[w] is not None
        assert len({a for a, pc in zip(arrived, pcs) if pc is not None}) <= 1, "wavefronts wait at different barriers"
    assert len(set(arrived)) == 1, "strands passed different numbers of barriers"
    if prio_logs is not None:
        prio_logs.extend(ln.prio_log for ln in lanes)
    outs = []
    for s in range(n_outputs):
        limbs = [out_region.words.get(OUT_BASE + 4 * ((s * 8 + l) * n_vec + vec)) for l in range(8)]
        outs.append(None if None in limbs else F.from_limbs32(limbs))
    unsat = sum(ln.operand_written["%[unsat]"] for ln in lanes)
    first = min(ln.operand_written["%[first]"] for ln in lanes)
    return outs, unsat, first


def test_priority_turns_of_a_strand_tape():
    """With turns on, every barrier sets the wavefront's priority to ((clock >> shift) + phase) & 3: the workgroups of
    a CU (phases 0..3) hold four different priorities at any time, each priority for a quarter of the time; results do
    not change; without turns the stream never touches the priority."""
    import zkir_amd as z
    o = z.default_lower_options()
    o.strands = 4
    t = z.Lowerer(z.poseidon_style_gadget(4, 6, 2), o).lower()
    ops, consts, so = t.ops(), t.consts(), list(t.hdr.strand_ops)[:4]
    rnd = random.Random(5)
    inputs = [[rnd.randrange(F.P) for _ in range(4)] for _ in range(2)]
    want, _, _ = F.eval_tape_one(ops, consts, inputs[1], strand_ops=so)
    n_barriers = sum(1 for op in ops[:so[0]] if op[0] & 0xFF == F.OP_BARRIER)
    assert n_barriers >= 4
    per_phase = []
    for phase in (0, 1, 2, 3, 0xFFFFFFFF):
        logs = []
        outs, unsat, _ = run_strands(ops, so, consts, inputs, t.hdr.n_outputs, t.hdr.n_lslots, t.hdr.n_gslots, t.hdr.n_xslots,
                                     1, 9, rot=phase, clock_step=1 << 12, prio_logs=logs)
        assert outs == [want[k] for k in range(t.hdr.n_outputs)] and unsat == 0
        if phase == 0xFFFFFFFF:
            assert all(log == [0] for log in logs)  # only the reset at the end of the stream
        else:
            # the clock advances one turn per barrier in this run: barrier k sets (k + phase) & 3; then the final reset
            assert all(log == [(k + phase) & 3 for k in range(n_barriers)] + [0] for log in logs)
            per_phase.append(logs[0][:n_barriers])
    for k in range(n_barriers):
        assert sorted(p[k] for p in per_phase) == [0, 1, 2, 3]


@pytest.mark.parametrize("n_strands", [4, 8])
def test_strand_tapes_on_the_model(n_strands):
    """Gadget and generated circuits lowered into strand tapes: the stream, run for one lane of each of
    the four wavefronts with LDS shared between them, gives the big-integer interpreter's outputs;
    a planted wrong `c` operand is reported under the constraint number its CHECK op carries."""
    import zkir_amd as z
    o = z.default_lower_options()
    o.strands = n_strands
    rnd = random.Random(21)
    g = z.default_generator_config()
    g.forced_inputs, g.inputs_readable, g.arithmetic_only, g.quadratic_aware, g.no_loops, g.safe_array_index = 4, 1, 1, 1, 1, 1
    g.min_instructions = g.max_instructions = 30
    circuits = [(z.poseidon_style_gadget(4, 3, 2), 4), (z.poseidon_style_gadget(2, 4, 3), 2),
                (z.Generator(3, g).generate(), 4), (z.Generator(8, g).generate(), 4)]
    for circuit, n_in in circuits:
        t = z.Lowerer(circuit, o).lower()
        assert t.hdr.flags & 1
        ops, consts, so = t.ops(), t.consts(), list(t.hdr.strand_ops)[:t.hdr.n_strands]
        assert len(so) == n_strands
        inputs = [[rnd.randrange(F.P) for _ in range(n_in)] for _ in range(3)]
        for vec, lane in ((0, 0), (2, 37)):
            want, want_unsat, _ = F.eval_tape_one(ops, consts, inputs[vec], strand_ops=so)
            outs, unsat, first = run_strands(ops, so, consts, inputs, t.hdr.n_outputs, t.hdr.n_lslots, t.hdr.n_gslots,
                                             t.hdr.n_xslots, vec, lane)
            assert outs == [want[k] for k in range(t.hdr.n_outputs)] and unsat == 0 and first == 0xFFFFFFFF
        # plant a fault: constraint 5 compares with constant 0
        bad = [list(op) for op in ops]
        for op in bad:
            if op[0] & 0xFF == F.OP_CHECK and (op[0] >> 8) - 1 == 5:
                op[1] = O(F.K_CONST, 0)
        want, want_unsat, _ = F.eval_tape_one(bad, consts, inputs[1], strand_ops=so)
        outs, unsat, first = run_strands(bad, so, consts, inputs, t.hdr.n_outputs, t.hdr.n_lslots, t.hdr.n_gslots,
                                         t.hdr.n_xslots, 1, 5)
        assert want_unsat == [5] and unsat == 1 and first == 5
