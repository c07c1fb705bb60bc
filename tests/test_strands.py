"""Strand tapes: four op streams of one circuit, run side by side by the four wavefronts of a
workgroup on the same 64 vectors, meeting in exchange slots across barriers
(include/zkir_hip.h, "Strand tapes").  CPU tests cover the format, the checkers and the
lowering; the gpu tests compare the kernel with the C checker bit for bit."""
import random
import struct

import numpy as np
import pytest

import zkir_amd as z
from oracle import c_oracle as C
from oracle import fr_ref as F
from zkir_amd import api

G = lambda i: F.operand(F.K_GSLOT, i)  # noqa: E731
L = lambda i: F.operand(F.K_LSLOT, i)  # noqa: E731
K = lambda i: F.operand(F.K_CONST, i)  # noqa: E731
I = lambda i: F.operand(F.K_INPUT, i)  # noqa: E731
X = lambda i: F.operand(F.K_XSLOT, i)  # noqa: E731
ACC, NONE = F.operand(F.K_ACC), F.operand(F.K_NONE)
CHECK = lambda k: F.OP_CHECK | ((k + 1) << 8)  # noqa: E731
BAR = (F.OP_BARRIER, NONE, NONE, NONE)


def strand_tape(strands, consts, n_inputs, n_outputs, n_g, n_l, n_x):
    ops = [o for s in strands for o in s]
    checks = sorted((o[0] >> 8, o) for o in ops if o[0] & 0xFF == F.OP_CHECK)
    hdr = api.TapeHdr(0x31544B5A, 2, n_inputs, n_outputs, len(consts), len(ops), n_g, n_l, len(checks), 0, 1,
                      (ctypes_u32x4(*[len(s) for s in strands])), n_x, len(strands), 0)
    blob = b"".join(struct.pack("<4I", *o) for o in ops) + b"".join(struct.pack("<8I", *F.limbs32(c)) for c in consts)
    blob += b"".join(struct.pack("<3I", o[2], o[3], o[1]) for _, o in checks)
    return hdr, blob


def ctypes_u32x4(*counts):
    """The header's strand_ops field: eight entries, the unused ones zero."""
    import ctypes
    return (ctypes.c_uint32 * 8)(*counts)


def hand_strands(bad_check=False):
    """in0, in1 -> out0 = in0*in1, out1 = in0+in1, out2 = (in0-in1)^2, out3 = 7*out0; three constraints."""
    s0 = [(F.OP_INPUT, X(0), I(0), NONE), BAR,
          (F.OP_MUL, X(2), X(0), X(1)), (F.OP_OUTPUT, 0, ACC, NONE), (CHECK(0), ACC, X(0), X(1)), BAR]
    s1 = [(F.OP_INPUT, X(1), I(1), NONE), BAR,
          (F.OP_ADD, NONE, X(0), X(1)), (F.OP_OUTPUT, 1, ACC, NONE), BAR]
    s2 = [BAR, (F.OP_SUB, G(0), X(0), X(1)), (F.OP_MUL, L(0), ACC, ACC), (F.OP_OUTPUT, 2, L(0), NONE),
          (CHECK(1), L(0), G(0), G(0)), BAR]
    s3 = [BAR, BAR, (F.OP_MUL, NONE, X(2), K(0)), (F.OP_OUTPUT, 3, ACC, NONE),
          (CHECK(2), X(0) if bad_check else ACC, X(2), K(0))]
    return strand_tape([s0, s1, s2, s3], [7], 2, 4, 1, 1, 3)


def test_checkers_agree_on_a_hand_written_strand_tape():
    rnd = random.Random(5)
    hdr, blob = hand_strands()
    ops = [struct.unpack_from("<4I", blob, 16 * i) for i in range(hdr.n_ops)]
    vecs = [[rnd.randrange(F.P) for _ in range(2)] for _ in range(9)] + [[0, 0], [F.P - 1, 1]]
    inp = z.pack_inputs(vecs, 2)
    out, unsat, first, faults = C.eval_cpu(bytes(hdr), blob, inp, len(vecs), 4)
    got = z.unpack_outputs(out)
    for t, (a, b) in enumerate(vecs):
        want = [a * b % F.P, (a + b) % F.P, (a - b) ** 2 % F.P, 7 * a * b % F.P]
        assert got[t] == want
        o_py, u_py, f_py = F.eval_tape_one(ops, [7], [a, b], strand_ops=list(hdr.strand_ops)[:hdr.n_strands])
        assert [o_py[k] for k in range(4)] == want and u_py == [] and f_py == 0
    assert (unsat, first, faults) == (0, None, 0)
    # a constraint that does not hold is reported under the number its CHECK op carries
    hdr, blob = hand_strands(bad_check=True)
    ops = [struct.unpack_from("<4I", blob, 16 * i) for i in range(hdr.n_ops)]
    out, unsat, first, faults = C.eval_cpu(bytes(hdr), blob, inp, len(vecs), 4)
    bad = [t for t, (a, b) in enumerate(vecs) if 7 * a * b % F.P != a]
    assert unsat == len(bad) and first == (bad[0], 2)
    assert F.eval_tape_one(ops, [7], vecs[bad[0]], strand_ops=list(hdr.strand_ops)[:hdr.n_strands])[1] == [2]


def test_validation_of_strand_tapes():
    hdr, blob = hand_strands()
    z.Tape.from_parts(hdr, blob)  # well formed
    ops = [list(struct.unpack_from("<4I", blob, 16 * i)) for i in range(hdr.n_ops)]
    tail = blob[16 * hdr.n_ops:]

    def rebuilt(ops2, **hdr_changes):
        h = api.TapeHdr.from_buffer_copy(bytes(hdr))
        for k, v in hdr_changes.items():
            setattr(h, k, v)
        return h, b"".join(struct.pack("<4I", *o) for o in ops2) + tail

    cases = []
    o2 = [list(o) for o in ops]
    o2[1] = [F.OP_NOP, NONE, NONE, NONE]  # strand 0 loses a barrier: the wavefronts would wait forever
    cases.append(rebuilt(o2))
    o2 = [list(o) for o in ops]
    o2[2][1] = X(3)  # exchange slot beyond n_xslots
    cases.append(rebuilt(o2))
    o2 = [list(o) for o in ops]
    o2[4][0] = F.OP_CHECK  # no constraint number
    cases.append(rebuilt(o2))
    o2 = [list(o) for o in ops]
    o2[4][0] = CHECK(1)  # number used twice
    cases.append(rebuilt(o2))
    cases.append(rebuilt(ops, strand_ops=ctypes_u32x4(6, 5, 6, 4)))  # counts do not add up
    cases.append(rebuilt(ops, flags=0))  # strand fields / exchange operands on a plain tape
    cases.append(rebuilt(ops, n_xslots=49))
    cases.append(rebuilt(ops, n_strands=5))  # a strand tape has 4 or 8 strands
    cases.append(rebuilt(ops, n_strands=8))  # the fifth to eighth hold no barriers
    cases.append(rebuilt(ops, n_scale_ops=99))  # more scaling products than products
    for h, b in cases:
        with pytest.raises(z.ZkirError):
            z.Tape.from_parts(h, b)


@pytest.mark.gpu
def test_hand_written_strand_tape_on_the_gpu():
    rnd = random.Random(6)
    ctx = z.Context(0)
    try:
        for bad, n in ((False, 1), (False, 64), (False, 65), (True, 200), (False, 1000)):
            hdr, blob = hand_strands(bad_check=bad)
            tape = z.Tape.from_parts(hdr, blob)
            vecs = [[rnd.randrange(F.P) for _ in range(2)] for _ in range(n)]
            inp = z.pack_inputs(vecs, 2)
            tid = ctx.upload(tape)
            out, unsat, first, faults = ctx.eval_batch(tid, inp, n)
            ctx.release(tid)
            o_c, u_c, first_c, f_c = C.eval_cpu(bytes(hdr), blob, inp, n, 4)
            assert np.array_equal(out, o_c)
            assert (unsat, first, faults) == (u_c, first_c, f_c)
            assert (unsat > 0) == bad
    finally:
        ctx.close()


@pytest.mark.gpu
def test_a_launch_beyond_the_lds_of_a_cu_is_refused_with_a_message():
    """A hand-written eight-strand tape that declares 16 window rows and 48 exchange slots would need 397 KiB of LDS
    per workgroup: the evaluation call returns ZKIR_E_LIMIT and says why, nothing is launched."""
    s = [[(F.OP_INPUT, X(k % 2), I(k % 2), NONE), BAR, (F.OP_OUTPUT, k % 4, X(0), NONE)] if k < 2 else
         [BAR, (F.OP_OUTPUT, k % 4, X(1), NONE)] for k in range(8)]
    hdr, blob = strand_tape(s, [7], 2, 4, 1, 16, 48)
    tape = z.Tape.from_parts(hdr, blob)  # the format allows it
    ctx = z.Context(0)
    try:
        tid = ctx.upload(tape)
        inp = z.pack_inputs([[3, 4]] * 5, 2)
        with pytest.raises(z.ZkirError) as e:
            ctx.eval_batch(tid, inp, 5)
        assert e.value.code == z.E_LIMIT and "LDS" in str(e.value)
        ctx.release(tid)
    finally:
        ctx.close()


def _ext_config(n, seed_inputs=6):
    g = z.default_generator_config()
    g.min_instructions = g.max_instructions = n
    g.forced_inputs, g.inputs_readable, g.arithmetic_only, g.quadratic_aware = seed_inputs, 1, 1, 1
    g.no_loops, g.safe_array_index = 1, 1
    return g


def _strand_options(**kw):
    o = z.default_lower_options()
    o.strands = 4
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _circuits():
    yield "gadget width 4", z.poseidon_style_gadget(4, 12, 3), 4
    yield "gadget width 3", z.poseidon_style_gadget(3, 9, 4), 3
    yield "gadget width 1", z.poseidon_style_gadget(1, 9, 5), 1
    for seed in range(1, 25):
        yield "generated %d" % seed, z.Generator(seed, _ext_config(20 + 7 * seed)).generate(), 6


@pytest.mark.parametrize("n_strands", [4, 8])
def test_lowering_into_strands_keeps_every_result(n_strands):
    """The strand tape of a circuit gives, vector for vector, the outputs, the unsatisfied count and
    the first unsatisfied pair of its plain tape (C checker), and its ops are the plain tape's ops
    plus barriers."""
    rnd = random.Random(9)
    n_strand_tapes = 0
    for name, circuit, n_in in _circuits():
        plain = z.Lowerer(circuit).lower()
        st = z.Lowerer(circuit, _strand_options(strands=n_strands)).lower()
        vecs = [[rnd.randrange(F.P) for _ in range(n_in)] for _ in range(11)]
        inp = z.pack_inputs(vecs, n_in)
        want = C.eval_cpu(bytes(plain.hdr), plain.blob(), inp, len(vecs), plain.hdr.n_outputs)
        got = C.eval_cpu(bytes(st.hdr), st.blob(), inp, len(vecs), st.hdr.n_outputs)
        assert np.array_equal(want[0], got[0]) and want[1:] == got[1:], name
        if st.hdr.flags & 1:
            n_strand_tapes += 1
            ops = st.ops()
            n_bar = sum(1 for o in ops if o[0] & 0xFF == F.OP_BARRIER)
            assert st.hdr.n_strands == n_strands and list(st.hdr.strand_ops)[n_strands:] == [0] * (8 - n_strands)
            in_order = z.default_lower_options()
            in_order.reorder = in_order.window_cache = 0  # (a plain tape otherwise holds COPY ops for values with two homes)
            assert n_bar % n_strands == 0 and len(ops) - n_bar == z.Lowerer(circuit, in_order).lower().hdr.n_ops, name
            assert (st.hdr.n_constraints, st.hdr.n_mul_ops) == (plain.hdr.n_constraints, plain.hdr.n_mul_ops)
            assert st.hdr.n_xslots <= (32 if n_strands == 4 else 48)
            # whatever window the caller asks for, no mix of strand tapes in one launch may exceed a CU's LDS (the
            # launch takes the most rows and the most exchange slots among its tapes; 9 KiB rows in the 29-bit kernel)
            greedy = z.Lowerer(circuit, _strand_options(strands=n_strands, n_lslots=16)).lower().hdr
            assert 2048 + (2 if n_strands == 8 else 1) * max(greedy.n_lslots, 4 if n_strands == 4 else 2) * 9216 \
                + (32 if n_strands == 4 else 48) * 2304 + 16 <= 160 * 1024 and greedy.n_lslots <= (4 if n_strands == 4 else 2)
            if n_strands == 8:  # strands 4..7 hold only what defines nothing
                tail = ops[sum(list(st.hdr.strand_ops)[:4]):]
                assert all(o[0] & 0xFF in (F.OP_OUTPUT, F.OP_OUTPUT_SCALED, F.OP_CHECK, F.OP_BARRIER) for o in tail), name
            # the Python interpreter agrees on one vector
            o_py, u_py, _ = F.eval_tape_one(ops, st.consts(), vecs[0], strand_ops=list(st.hdr.strand_ops)[:st.hdr.n_strands])
            assert [o_py[k] for k in range(st.hdr.n_outputs)] == z.unpack_outputs(got[0])[0] and u_py == []
        else:
            assert st.blob() == plain.blob(), name  # fell back: the plain tape, bit for bit
    assert n_strand_tapes >= (10 if n_strands == 4 else 6)


@pytest.mark.parametrize("n_strands", [4, 8])
def test_a_broken_constraint_keeps_its_number_in_a_strand_tape(n_strands):
    circuit = z.poseidon_style_gadget(4, 6, 3)
    plain = z.Lowerer(circuit).lower()
    st = z.Lowerer(circuit, _strand_options(strands=n_strands)).lower()
    assert st.hdr.flags & 1
    rnd = random.Random(2)
    vecs = [[rnd.randrange(F.P) for _ in range(4)] for _ in range(5)]
    inp = z.pack_inputs(vecs, 4)
    # swap the c operand of constraint 17 for a constant in both tapes
    def broken(t):
        ops = [list(o) for o in t.ops()]
        k = 0
        for o in ops:
            if o[0] & 0xFF == F.OP_CHECK:
                idx = (o[0] >> 8) - 1 if t.hdr.flags & 1 else k
                k += 1
                if idx == 17:
                    o[1] = K(0)
        blob = b"".join(struct.pack("<4I", *o) for o in ops) + t.blob()[16 * t.hdr.n_ops:]
        return C.eval_cpu(bytes(t.hdr), blob, inp, len(vecs), t.hdr.n_outputs)
    a, b = broken(plain), broken(st)
    assert a[1] == b[1] == len(vecs) and a[2] == b[2] == (0, 17)


def test_strand_lowering_refuses_nothing_and_falls_back():
    # operators beyond + - *: plain tape; exchange slots beyond the cap: plain tape; strands must be 0 or 4
    c = z.Generator(5).generate()  # reference-shaped: every operator
    try:
        t = z.Lowerer(c, _strand_options()).lower()
        assert t.hdr.flags == 0
    except z.LowerRejected:
        pass
    big = z.Generator(7, _ext_config(900)).generate()
    assert z.Lowerer(big, _strand_options()).lower().hdr.flags == 0
    assert z.Lowerer(z.poseidon_style_gadget(4, 5, 1), _strand_options(max_xslots=2)).lower().hdr.flags == 0
    with pytest.raises(z.ZkirError):
        z.Lowerer(z.poseidon_style_gadget(4, 5, 1), _strand_options(strands=3)).lower()


@pytest.mark.gpu
@pytest.mark.parametrize("n_strands", [4, 8])
def test_lowered_strand_tapes_on_the_gpu(n_strands):
    rnd = random.Random(10)
    ctx = z.Context(0)
    try:
        n_strand_tapes = 0
        for name, circuit, n_in in _circuits():
            st = z.Lowerer(circuit, _strand_options(strands=n_strands)).lower()
            n_strand_tapes += st.hdr.flags & 1
            n = rnd.choice([1, 63, 64, 65, 200, 777])
            vecs = [[rnd.randrange(F.P) for _ in range(n_in)] for _ in range(n)]
            inp = z.pack_inputs(vecs, n_in)
            tid = ctx.upload(st)
            out, unsat, first, faults = ctx.eval_batch(tid, inp, n)
            ctx.release(tid)
            o_c, u_c, first_c, f_c = C.eval_cpu(bytes(st.hdr), st.blob(), inp, n, st.hdr.n_outputs)
            assert np.array_equal(out, o_c), name
            assert (unsat, first, faults) == (u_c, first_c, f_c) == (0, None, 0), name
        assert n_strand_tapes >= (10 if n_strands == 4 else 6)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_strand_and_plain_tapes_in_one_call():
    """zkir_eval_multi over strand tapes, over plain tapes and over both kinds mixed (two launches behind
    one call): per-tape counts in the caller's order, a planted fault found in either kind."""
    ctx = z.Context(0)
    try:
        circuits = [z.poseidon_style_gadget(4, 3 + k, k + 1) for k in range(6)]
        strand = [z.Lowerer(c, _strand_options()).lower() for c in circuits]
        plain = [z.Lowerer(c).lower() for c in circuits]
        strand8 = [z.Lowerer(c, _strand_options(strands=8)).lower() for c in circuits]
        assert all(t.hdr.flags & 1 for t in strand) and not any(t.hdr.flags & 1 for t in plain)
        assert all(t.hdr.n_strands == 8 for t in strand8)

        def broken(t):  # constraint 2 compared with a constant: fails for every vector
            ops = [list(o) for o in t.ops()]
            k = 0
            for o in ops:
                if o[0] & 0xFF == F.OP_CHECK:
                    idx = (o[0] >> 8) - 1 if t.hdr.flags & 1 else k
                    k += 1
                    if idx == 2:
                        o[1] = K(0)
            blob = b"".join(struct.pack("<4I", *o) for o in ops) + t.blob()[16 * t.hdr.n_ops:]
            return z.Tape.from_parts(t.hdr, blob)

        n = 130
        rnd = random.Random(3)
        mixed2 = [strand[0], plain[1], broken(strand[2]), plain[3], strand[4], broken(plain[5])]
        mixed3 = [strand8[0], plain[1], broken(strand[2]), strand8[3], strand[4], broken(strand8[5]), broken(plain[0])]  # three launches
        for tapes in (strand, plain, strand8, mixed2, mixed3):
            tids = ctx.upload_many(tapes)
            blocks, want = [], []
            for t in tapes:
                vecs = [[rnd.randrange(F.P) for _ in range(4)] for _ in range(n)]
                inp = z.pack_inputs(vecs, 4)
                blocks.append(inp.reshape(-1))
                want.append(C.eval_cpu(bytes(t.hdr), t.blob(), inp, n, t.hdr.n_outputs, want_outputs=False)[1])
            unsat, faults = ctx.eval_multi(tids, np.concatenate(blocks), n)
            assert list(unsat) == want and faults == 0
            assert sum(1 for u in unsat if u) == (2 if tapes is mixed2 else 3 if tapes is mixed3 else 0)
            for tid in tids:
                ctx.release(tid)
    finally:
        ctx.close()
