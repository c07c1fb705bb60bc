"""Parity of the gfx950 kernels, through the C ABI, against the checkers of oracle/
(bit-exact: the path is integer arithmetic).  Needs a real MI355X."""
import json
import os
import random
import struct

import numpy as np
import pytest

import zkir_amd as z
from oracle import c_oracle as C
from oracle import fr_ref as F
from oracle import zkir_ref as R
from zkir_amd import api

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

G = lambda i: F.operand(F.K_GSLOT, i)  # noqa: E731
L = lambda i: F.operand(F.K_LSLOT, i)  # noqa: E731
K = lambda i: F.operand(F.K_CONST, i)  # noqa: E731
I = lambda i: F.operand(F.K_INPUT, i)  # noqa: E731
ACC, NONE = F.operand(F.K_ACC), F.operand(F.K_NONE)


@pytest.fixture(scope="module")
def ctx():
    c = z.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True, params=["generated-stream", "stream32", "compiled"])
def interpreter_kernel(request):
    """Every test runs three times: + - * tapes through the generated instruction stream on 9 x 29-bit limbs
    (ZKIR_INTERP=lean29), through the generated stream on 8 x 32-bit limbs (ZKIR_INTERP=lean32) and through the
    compiled kernel (ZKIR_INTERP=compiled); tapes with other operators take the compiled every-operator kernel
    every time.  (Without the variable the library picks between the two streams by the tape's device slots.)"""
    old = os.environ.get("ZKIR_INTERP")
    os.environ["ZKIR_INTERP"] = {"compiled": "compiled", "stream32": "lean32", "generated-stream": "lean29"}[request.param]
    yield request.param
    if old is None:
        os.environ.pop("ZKIR_INTERP", None)
    else:
        os.environ["ZKIR_INTERP"] = old


def hand_tape(ops, consts, n_inputs, n_outputs, n_g, n_l):
    n_con = sum(1 for o in ops if o[0] == F.OP_CHECK)
    hdr = api.TapeHdr(0x31544B5A, 2, n_inputs, n_outputs, len(consts), len(ops), n_g, n_l, n_con, 0, 0)
    blob = b"".join(struct.pack("<4I", *o) for o in ops) + b"".join(struct.pack("<8I", *F.limbs32(c)) for c in consts)
    blob += b"".join(struct.pack("<3I", o[2], o[3], o[1]) for o in ops if o[0] == F.OP_CHECK)
    return z.Tape.from_parts(hdr, blob)


def ext_config(**kw):
    g = z.default_generator_config()
    for k, v in kw.items():
        setattr(g, k, v)
    return g


def random_vectors(rnd, n_inputs, n, special_share=0.25):
    special = [0, 1, 2, F.P - 1, F.P - 2, F.HALF, F.HALF + 1, (1 << 64) - 1, 1 << 253, 253, 254, 255, F.P - 253, F.P - 254]
    return [[rnd.choice(special) if rnd.random() < special_share else rnd.randrange(F.P) for _ in range(n_inputs)]
            for _ in range(n)]


def check_against_c_oracle(ctx, tape, vecs):
    n = len(vecs)
    inp = z.pack_inputs(vecs, tape.hdr.n_inputs)
    tid = ctx.upload(tape)
    out, unsat, first, faults = ctx.eval_batch(tid, inp, n)
    ctx.release(tid)
    o_c, u_c, first_c, f_c = C.eval_cpu(bytes(tape.hdr), tape.blob(), inp, n, tape.hdr.n_outputs)
    assert np.array_equal(out, o_c)
    assert (unsat, first, faults) == (u_c, first_c, f_c)
    return out, unsat, first, faults


def test_every_operator_against_known_answers(ctx):
    with open(os.path.join(GOLD, "fr_kat.json")) as f:
        kat = json.load(f)
    for name, ent in kat.items():
        code, rows = ent["code"], ent["rows"]
        ops = [(F.OP_INPUT, G(0), I(0), NONE), (F.OP_INPUT, L(0), I(1), NONE),
               (code, G(1), G(0), L(0) if code >= F.OP_ADD else NONE), (F.OP_OUTPUT, 0, G(1), NONE)]
        tape = hand_tape(ops, [], 2, 1, 2, 1)
        vecs = [[int(a, 16), int(b, 16)] for a, b, _, _ in rows]
        tid = ctx.upload(tape)
        out, unsat, first, faults = ctx.eval_batch(tid, z.pack_inputs(vecs, 2), len(vecs))
        ctx.release(tid)
        got = [v[0] for v in z.unpack_outputs(out)]
        assert got == [int(r, 16) for _, _, r, _ in rows], name
        assert faults == sum(f for _, _, _, f in rows), name
        assert unsat == 0 and first is None


def test_operand_kinds_and_accumulator(ctx):
    """Every source / destination kind of the tape format, mixed in one tape."""
    ops = [(F.OP_INPUT, G(0), I(0), NONE), (F.OP_INPUT, L(0), I(1), NONE),
           (F.OP_MUL, G(1), G(0), L(0)), (F.OP_OUTPUT, 0, G(1), NONE),
           (F.OP_ADD, L(1), G(0), L(0)), (F.OP_OUTPUT, 1, ACC, NONE),
           (F.OP_SUB, NONE, G(0), L(0)), (F.OP_OUTPUT, 2, ACC, NONE),
           (F.OP_MUL, G(2), ACC, K(0)), (F.OP_OUTPUT, 3, G(2), NONE),
           (F.OP_CHECK, G(1), G(0), L(0)),  # holds
           (F.OP_CHECK, G(2), G(0), L(0)),  # almost never holds
           (F.OP_NEG, G(3), G(0), NONE), (F.OP_OUTPUT, 4, G(3), NONE),
           (F.OP_COPY, L(2), K(1), NONE), (F.OP_OUTPUT, 5, L(2), NONE),
           (F.OP_NOP, 0, 0, 0),
           (F.OP_MUL, NONE, K(1), K(0)), (F.OP_CHECK, ACC, K(1), K(0))]
    tape = hand_tape(ops, [12345678901234567890123456789 % F.P, F.P - 1], 2, 6, 4, 3)
    vecs = random_vectors(random.Random(1), 2, 1000)
    out, unsat, first, faults = check_against_c_oracle(ctx, tape, vecs)
    for t, v in enumerate(vecs[:50]):
        o, u, _ = F.eval_tape_one(ops, tape.consts(), v)
        assert z.unpack_outputs(out[:, :, t:t + 1])[0] == [o[k] for k in range(6)]
    assert unsat >= 990 and first[0] == 0 and first[1] == 1


def test_injected_fault_is_counted_and_located(ctx):
    """c = a*b for every vector, then one limb of c is flipped in a few vectors."""
    ops = [(F.OP_INPUT, G(0), I(0), NONE), (F.OP_INPUT, G(1), I(1), NONE), (F.OP_INPUT, G(2), I(2), NONE),
           (F.OP_CHECK, G(2), G(0), G(1)), (F.OP_CHECK, G(2), G(1), G(0))]
    tape = hand_tape(ops, [], 3, 0, 3, 0)
    rnd = random.Random(2)
    n = 5000
    vecs = [[a, b, F.mul(a, b)] for a, b in ((rnd.randrange(F.P), rnd.randrange(F.P)) for _ in range(n))]
    tid = ctx.upload(tape)
    inp = z.pack_inputs(vecs, 3)
    _, unsat, first, _ = ctx.eval_batch(tid, inp, n, want_outputs=False)
    assert unsat == 0 and first is None
    flipped = sorted(rnd.sample(range(n), 7))
    for t in flipped:
        inp[2, rnd.randrange(8), t] ^= 1 << rnd.randrange(29)  # stays below p: limb 7 keeps its top bits
    _, unsat, first, _ = ctx.eval_batch(tid, inp, n, want_outputs=False)
    assert unsat == 2 * len(flipped) and first == (flipped[0], 0)
    ctx.release(tid)


def test_checks_against_the_constant_one(ctx):
    """`value * 1 == signal` (the constraint of a signal with a linear value): the kernels compare the left factor
    as it is when the right factor is the pool's constant 1.  Every operand kind as left factor, satisfied vectors
    and a few with one bit of the signal flipped, against the C checker (which multiplies by one)."""
    ops = [(F.OP_INPUT, G(0), I(0), NONE), (F.OP_INPUT, L(0), I(1), NONE), (F.OP_INPUT, G(1), I(2), NONE),
           (F.OP_CHECK, G(1), G(0), K(1)),              # device slot * 1 == device slot
           (F.OP_CHECK, G(1), L(0), K(1)),              # LDS slot * 1
           (F.OP_ADD, G(2), G(0), K(0)), (F.OP_CHECK, ACC, ACC, K(1)), (F.OP_CHECK, G(2), ACC, K(1)),
           (F.OP_CHECK, ACC, G(2), K(1)),
           (F.OP_CHECK, K(2), K(2), K(1)),              # a constant against itself
           (F.OP_CHECK, G(1), G(0), K(2)),              # the same check against p - 1: a product, fails unless x = 0
           (F.OP_MUL, G(3), G(0), K(1)), (F.OP_CHECK, G(1), G(3), K(1)), (F.OP_OUTPUT, 0, G(3), NONE)]
    tape = hand_tape(ops, [77, 1, F.P - 1], 3, 1, 4, 1)
    rnd = random.Random(31)
    n = 3000
    vecs = [[x, x, x] for x in (rnd.choice([0, 1, F.P - 1]) if rnd.random() < 0.1 else rnd.randrange(F.P) for _ in range(n))]
    inp = z.pack_inputs(vecs, 3)
    tid = ctx.upload(tape)
    out, unsat, first, faults = ctx.eval_batch(tid, inp, n)
    o_c, u_c, first_c, f_c = C.eval_cpu(bytes(tape.hdr), tape.blob(), inp, n, 1)
    zeros = sum(1 for v in vecs if v[0] == 0)
    assert np.array_equal(out, o_c) and (unsat, first, faults) == (u_c, first_c, f_c) and unsat == n - zeros
    flipped = sorted(rnd.sample(range(n), 9))
    for t in flipped:
        inp[2, rnd.randrange(7), t] ^= 1 << rnd.randrange(32)  # limb 7 untouched: the value stays below p
    out, unsat, first, faults = ctx.eval_batch(tid, inp, n)
    o_c, u_c, first_c, f_c = C.eval_cpu(bytes(tape.hdr), tape.blob(), inp, n, 1)
    assert np.array_equal(out, o_c) and (unsat, first, faults) == (u_c, first_c, f_c)
    assert unsat >= n - zeros - 9 + 3 * len(flipped)
    ctx.release(tid)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 1023])
def test_batch_sizes_around_the_workgroup(ctx, n):
    c = z.Generator(11, ext_config(min_instructions=80, max_instructions=80, forced_inputs=3, inputs_readable=1,
                                   arithmetic_only=1, quadratic_aware=1, no_loops=1, safe_array_index=1)).generate()
    tape = z.Lowerer(c).lower()
    check_against_c_oracle(ctx, tape, random_vectors(random.Random(n), tape.hdr.n_inputs, n))


def test_empty_batch_and_tape_without_inputs(ctx):
    txt = '(circuit (vars (var 0 "o" field output (init 0))) (instrs (decl 0) (constrain 0 (pow (const "3") (const "200")))))'
    o = z.default_lower_options()
    o.fold_constants = 0
    tape = z.Lowerer(z.Circuit.parse(txt), o).lower()
    assert tape.hdr.n_inputs == 0
    tid = ctx.upload(tape)
    out, unsat, first, faults = ctx.eval_batch(tid, None, 0)
    assert out.shape == (1, 8, 0) and unsat == 0 and first is None
    out, unsat, first, faults = ctx.eval_batch(tid, None, 300)
    assert all(v == [pow(3, 200, F.P)] for v in z.unpack_outputs(out)) and unsat == 0
    ctx.release(tid)
    with pytest.raises(z.ZkirError):
        ctx.release(tid)  # already released: an error code, not a crash


def test_error_codes_not_crashes(ctx):
    """Misuse of the device entry points comes back as an error code with a message."""
    import ctypes
    lib = api.lib
    ops = [(F.OP_INPUT, G(0), I(0), NONE), (F.OP_OUTPUT, 0, G(0), NONE)]
    tape = hand_tape(ops, [], 1, 1, 1, 0)
    tid = ctx.upload(tape)
    unsat, first = ctypes.c_uint64(), (ctypes.c_uint32 * 2)()
    out = np.zeros((1, 8, 4), dtype=np.uint32)
    # unknown tape id
    rc = lib.zkir_eval_batch(ctx._h, 0x7FFFFFF0, None, 4, out.ctypes.data, ctypes.byref(unsat), first)
    assert rc == api.E_INVALID and b"unknown tape" in lib.zkir_last_error(ctx._h)
    # a tape with inputs, a batch, and no input pointer
    rc = lib.zkir_eval_batch(ctx._h, tid, None, 4, out.ctypes.data, ctypes.byref(unsat), first)
    assert rc == api.E_INVALID and b"inputs required" in lib.zkir_last_error(ctx._h)
    # null result pointers, null context
    assert lib.zkir_eval_batch(ctx._h, tid, None, 0, None, None, None) == api.E_INVALID
    assert lib.zkir_eval_batch(None, tid, None, 0, None, ctypes.byref(unsat), first) == api.E_INVALID
    assert lib.zkir_tape_upload(ctx._h, None, None, 0, None) == api.E_INVALID
    # a tape whose op names a slot beyond the header's count never reaches the device
    bad = bytearray(tape.blob())
    bad[4:8] = struct.pack("<I", G(5))  # dst of the first op
    tid2 = ctypes.c_uint32()
    rc = lib.zkir_tape_upload(ctx._h, ctypes.byref(tape.hdr), bytes(bad), len(bad), ctypes.byref(tid2))
    assert rc == api.E_INVALID and b"bad tape" in lib.zkir_last_error(ctx._h)
    # the context still works afterwards
    inp = z.pack_inputs([[7], [F.P - 1]], 1)
    got, u, f, faults = ctx.eval_batch(tid, inp, 2)
    assert z.unpack_outputs(got) == [[7], [F.P - 1]] and u == 0
    ctx.release(tid)
    # a device that does not exist
    c = ctypes.c_void_p()
    assert lib.zkir_ctx_create(1 << 20, ctypes.byref(c)) == api.E_INVALID and not c.value


def test_generated_circuits_against_c_oracle(ctx):
    n_ok = 0
    for seed in range(1, 40):
        cfg = dict(min_instructions=150, max_instructions=220, forced_inputs=5, inputs_readable=1, safe_array_index=1,
                   max_expr_depth=1 + seed % 3, arithmetic_only=seed % 2, quadratic_aware=(seed // 2) % 2,
                   no_loops=(seed // 4) % 2)
        c = z.Generator(seed, ext_config(**cfg)).generate()
        o = z.default_lower_options()
        o.fold_constants, o.max_iterations, o.n_lslots = seed % 2, 300, [0, 3, 8, 16][seed % 4]
        o.bnot_variant, o.use_acc, o.reuse_slots = (seed // 8) % 2, (seed // 3) % 2, (seed // 5) % 2
        try:
            tape = z.Lowerer(c, o).lower()
        except z.LowerRejected:
            continue
        n_ok += 1
        check_against_c_oracle(ctx, tape, random_vectors(random.Random(seed), tape.hdr.n_inputs, 300))
    assert n_ok >= 8


def test_reference_shaped_circuits_against_ast_interpreter(ctx):
    """Circuits exactly as src/generator.rs draws them, every constant expression
    evaluated on the GPU (fold_constants = 0), against the direct interpretation of the AST."""
    n_ok = 0
    for seed in range(1, 400):
        ref = R.Generator(R.Rng(seed)).generate()
        o = z.default_lower_options()
        o.fold_constants, o.max_iterations, o.max_ops = 0, 400, 20000
        try:
            tape = z.Lowerer(z.Circuit.parse(R.print_text(ref)), o).lower()
        except z.LowerRejected:
            continue
        n_ok += 1
        rnd = random.Random(seed)
        vecs = random_vectors(rnd, tape.hdr.n_inputs, 70)
        tid = ctx.upload(tape)
        out, unsat, first, faults = ctx.eval_batch(tid, z.pack_inputs(vecs, tape.hdr.n_inputs), len(vecs))
        ctx.release(tid)
        got = z.unpack_outputs(out)
        for t in (0, 33, 69):
            it = R.Interpreter(ref, vecs[t], max_iterations=400).run()
            assert got[t] == it.outputs and it.unsat() == []
        assert unsat == 0
    assert n_ok >= 5


def test_many_tapes_one_launch(ctx):
    rnd = random.Random(9)
    tapes, tids, blocks, want = [], [], [], []
    n = 130
    for k in range(14):
        if k % 5 == 4:  # a tape whose second constraint fails for some vectors
            ops = [(F.OP_INPUT, G(0), I(0), NONE), (F.OP_INPUT, G(1), I(1), NONE), (F.OP_MUL, G(2), G(0), G(1)),
                   (F.OP_CHECK, G(2), G(0), G(1)), (F.OP_CHECK, G(0), G(0), G(1))]
            tape = hand_tape(ops, [], 2, 0, 3, 0)
        else:
            cfg = dict(min_instructions=40 + 30 * k, max_instructions=40 + 30 * k, forced_inputs=2 + k % 3,
                       inputs_readable=1, arithmetic_only=1, quadratic_aware=1, no_loops=1, safe_array_index=1)
            o = z.default_lower_options()
            o.n_lslots = [0, 4, 16][k % 3]
            tape = z.Lowerer(z.Generator(100 + k, ext_config(**cfg)).generate(), o).lower()
        vecs = random_vectors(rnd, tape.hdr.n_inputs, n, special_share=0.5)
        inp = z.pack_inputs(vecs, tape.hdr.n_inputs)
        _, u_c, _, _ = C.eval_cpu(bytes(tape.hdr), tape.blob(), inp, n, tape.hdr.n_outputs, want_outputs=False)
        tapes.append(tape)
        tids.append(ctx.upload(tape))
        blocks.append(inp.reshape(-1))
        want.append(u_c)
    unsat, faults = ctx.eval_multi(tids, np.concatenate(blocks), n)
    assert unsat == want and any(want) and faults == 0
    assert ctx.eval_multi([], None, n) == ([], 0)
    for t in tids:
        ctx.release(t)


def test_pipelined_result_copy(ctx, monkeypatch):
    """Results of 128 MiB and more leave the device in pieces through pinned memory while host threads move
    the previous piece into the caller's buffer (zkir_eval_batch); ZKIR_COPY_PIPELINE=1 forces that path on
    small results.  Same bytes either way: a result of one partial piece, and one of several pieces."""
    rnd = random.Random(123)
    g = ext_config(min_instructions=160, max_instructions=160, forced_inputs=5, inputs_readable=1, arithmetic_only=1,
                   quadratic_aware=1, no_loops=1, safe_array_index=1)
    tape = z.Lowerer(z.Generator(9, g).generate()).lower()
    monkeypatch.setenv("ZKIR_COPY_PIPELINE", "1")
    for n in (1, 257, 1000):
        check_against_c_oracle(ctx, tape, random_vectors(rnd, tape.hdr.n_inputs, n))
    monkeypatch.delenv("ZKIR_COPY_PIPELINE")
    # several pieces of 64 MiB, the last one partial
    n = (150 << 20) // (32 * tape.hdr.n_outputs) + 77
    assert 32 * tape.hdr.n_outputs * n > (128 << 20)
    base = np.random.default_rng(5).integers(0, 1 << 32, size=(tape.hdr.n_inputs, 8, n), dtype=np.uint64).astype(np.uint32)
    base[:, 7, :] &= 0x0FFFFFFF
    tid = ctx.upload(tape)
    out, unsat, first, faults = ctx.eval_batch(tid, base, n)
    monkeypatch.setenv("ZKIR_COPY_PIPELINE", "0")
    out_plain, unsat_plain, _, _ = ctx.eval_batch(tid, base, n)
    ctx.release(tid)
    assert np.array_equal(out, out_plain) and (unsat, first, faults) == (0, None, 0) and unsat_plain == 0
    o_c, u_c, _, _ = C.eval_cpu(bytes(tape.hdr), tape.blob(), base, n, tape.hdr.n_outputs, threads=8)
    assert u_c == 0 and np.array_equal(out, o_c)


def test_batched_upload_gives_the_same_results(ctx):
    """zkir_tape_upload_many against one zkir_tape_upload per tape: plain, strand and every-operator tapes in
    one batch (evaluated one by one), then a multi-tape launch of the batch's plain tapes."""
    rnd = random.Random(77)
    so = z.default_lower_options()
    so.strands = 4
    tapes = [z.Lowerer(z.Generator(s, ext_config(min_instructions=30 + 9 * s, max_instructions=30 + 9 * s, forced_inputs=3,
                                                  inputs_readable=1, arithmetic_only=1, quadratic_aware=1, no_loops=1,
                                                  safe_array_index=1)).generate()).lower() for s in range(1, 9)]
    tapes.append(z.Lowerer(z.poseidon_style_gadget(4, 5, 2), so).lower())
    tapes.append(hand_tape([(F.OP_INPUT, G(0), I(0), NONE), (F.OP_LT, G(1), G(0), K(0)), (F.OP_OUTPUT, 0, G(1), NONE)],
                           [5], 1, 1, 2, 0))
    ids = ctx.upload_many(tapes)
    assert len(set(ids)) == len(tapes)
    n = 70
    for t, tid in zip(tapes, ids):
        vecs = random_vectors(rnd, t.hdr.n_inputs, n)
        inp = z.pack_inputs(vecs, t.hdr.n_inputs)
        got = ctx.eval_batch(tid, inp, n)
        single = ctx.upload(t)
        want = ctx.eval_batch(single, inp, n)
        ctx.release(single)
        o_c, u_c, first_c, f_c = C.eval_cpu(bytes(t.hdr), t.blob(), inp, n, t.hdr.n_outputs)
        assert np.array_equal(got[0], want[0]) and got[1:] == want[1:]
        assert np.array_equal(got[0], o_c) and got[1:] == (u_c, first_c, f_c)
    plain = [(t, tid) for t, tid in zip(tapes[:8], ids[:8])]
    blocks, want = [], []
    for t, _ in plain:
        inp = z.pack_inputs(random_vectors(rnd, t.hdr.n_inputs, n), t.hdr.n_inputs)
        blocks.append(inp.reshape(-1))
        want.append(C.eval_cpu(bytes(t.hdr), t.blob(), inp, n, t.hdr.n_outputs, want_outputs=False)[1])
    got, _ = ctx.eval_multi([tid for _, tid in plain], np.concatenate(blocks), n)
    assert got == want
    for tid in ids:
        ctx.release(tid)
    # a bad tape anywhere in the batch: nothing is uploaded
    bad_hdr = api.TapeHdr.from_buffer_copy(bytes(tapes[0].hdr))
    bad_hdr.n_consts += 1  # the blob no longer matches the header
    with pytest.raises(z.ZkirError):
        ctx.upload_many_parts([(tapes[1].hdr, tapes[1].blob()), (bad_hdr, tapes[0].blob())])
    # the same pairs with honest headers go through the validating entry point and run like the handles did
    pids = ctx.upload_many_parts([(t.hdr, t.blob()) for t, _ in plain])
    got2, _ = ctx.eval_multi(pids, np.concatenate(blocks), n)
    assert got2 == want
    for tid in pids:
        ctx.release(tid)


def test_campaign_on_one_gpu():
    """The campaign driver with the real evaluator: generated circuits satisfy their own
    constraints for every device-drawn vector, refused circuits are counted by class, and a
    hand-broken tape among them is reported as divergent."""
    from zkir_amd import campaign

    def config_of(cid):
        g = campaign.campaign_config(cid, min_constraints=8, max_constraints=120)
        if cid % 6 == 0:  # keep the reference's loops in some: lower refuses most of those
            g.no_loops, g.quadratic_aware = 0, 0
        return g

    o = z.default_lower_options()
    o.max_iterations = 200
    ev = campaign.GpuEvaluator(0, input_seed=5)
    res = campaign.run_local(range(40), 31, 300, ev, config_of=config_of, lower_options=o, group=7)
    tot = campaign.reduce_results(res)
    assert tot["circuits"] == 40 and tot["rejected"] == sum(tot["classes"][1:]) and 0 < tot["rejected"] < 40
    assert tot["unsat"] == 0 and tot["divergent"] == [] and tot["checks"] > 0

    def breaking(tapes, n_vectors, ids):  # turn one circuit's first product into a sum
        tapes = list(tapes)
        for k, cid in enumerate(ids):
            if cid == 13:
                ops = tapes[k].ops()
                j = next(i for i, op in enumerate(ops) if op[0] == F.OP_MUL)
                ops[j] = (F.OP_ADD,) + ops[j][1:]
                blob = tapes[k].blob()
                blob = b"".join(struct.pack("<4I", *op) for op in ops) + blob[16 * len(ops):]
                tapes[k] = z.Tape.from_parts(tapes[k].hdr, blob)
        return ev(tapes, n_vectors, ids)

    res = campaign.run_local(range(40), 31, 300, breaking, config_of=config_of, lower_options=o, group=7)
    assert res.divergent == [13] and res.unsat > 0
    ev.close()


def test_gadget_chain(ctx):
    c = z.poseidon_style_gadget(4, 60, 5)
    tape = z.Lowerer(c).lower()
    assert tape.hdr.n_constraints == 3 * 4 * 60
    vecs = random_vectors(random.Random(4), 4, 400)
    out, unsat, first, faults = check_against_c_oracle(ctx, tape, vecs)
    assert unsat == 0
    it = R.Interpreter(R.parse_text(c.text()), vecs[7]).run()
    assert z.unpack_outputs(out[:, :, 7:8])[0] == it.outputs


@pytest.mark.parametrize("signal_factors", [1, 0])
def test_device_pointer_entry_point_full_size(ctx, signal_factors):
    """The ~1k-constraint circuits of the benchmark (both factors of a constraint signals; a factor a constant one
    time in four, which lower turns into a tape of scaled values) over 2^20 vectors on device buffers: no constraint
    fails, sampled vectors match the checker on this tape and on the same circuit lowered plainly in program order,
    and permuting the batch permutes the outputs."""
    torch = pytest.importorskip("torch")
    cfg = dict(min_instructions=4200, max_instructions=4200, forced_inputs=16, inputs_readable=1, arithmetic_only=1,
               quadratic_aware=1, no_loops=1, safe_array_index=1, signal_factors=signal_factors, max_linear_terms=1)
    circuit = z.Generator(2024, ext_config(**cfg)).generate()
    tape = z.Lowerer(circuit).lower()
    h = tape.hdr
    assert 900 <= h.n_constraints <= 1200
    assert (h.n_scale_ops > 0) == (signal_factors == 0)  # the cost rule keeps scaled values only where they pay
    plain_options = z.default_lower_options()
    plain_options.reorder = plain_options.window_cache = plain_options.order_ops = plain_options.scaled_values = 0
    plain = z.Lowerer(circuit, plain_options).lower()
    assert plain.hdr.n_scale_ops == 0 and plain.hdr.n_outputs == h.n_outputs
    n = 1 << 20
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    inp = torch.randint(0, 1 << 32, (h.n_inputs, 8, n), dtype=torch.int64, device=dev, generator=gen).to(torch.int32)
    inp[:, 7, :] &= 0x0FFFFFFF  # every element below 2^252 < p
    out = torch.empty((h.n_outputs, 8, n), dtype=torch.int32, device=dev)
    res = torch.zeros(4, dtype=torch.int64, device=dev)
    tid = ctx.upload(tape)
    stream = torch.cuda.current_stream().cuda_stream
    ctx.eval_batch_dev(tid, inp.data_ptr(), n, out.data_ptr(), res.data_ptr(), stream)
    torch.cuda.synchronize()
    r = res.cpu().numpy().view(np.uint64)
    assert r[0] == 0 and r[1] == np.uint64(0xFFFFFFFFFFFFFFFF) and r[2] == 0
    # sampled vectors against the checker
    idx = torch.cat([torch.arange(0, 300), torch.arange(n // 2 - 150, n // 2 + 150), torch.arange(n - 300, n)]).to(dev)
    inp_s = inp[:, :, idx].cpu().numpy().view(np.uint32)
    out_s = out[:, :, idx].cpu().numpy().view(np.uint32)
    o_c, u_c, _, _ = C.eval_cpu(bytes(h), tape.blob(), np.ascontiguousarray(inp_s), len(idx), h.n_outputs)
    assert np.array_equal(np.ascontiguousarray(out_s), o_c) and u_c == 0
    o_p, u_p, _, _ = C.eval_cpu(bytes(plain.hdr), plain.blob(), np.ascontiguousarray(inp_s), len(idx), h.n_outputs)
    assert np.array_equal(o_p, o_c) and u_p == 0
    # order of the batch does not matter: reverse the vectors, outputs come out reversed
    inp_r = torch.flip(inp, dims=[2]).contiguous()
    out_r = torch.empty_like(out)
    ctx.eval_batch_dev(tid, inp_r.data_ptr(), n, out_r.data_ptr(), res.data_ptr(), stream)
    torch.cuda.synchronize()
    assert torch.equal(torch.flip(out_r, dims=[2]), out)
    ctx.release(tid)


def test_many_distinct_circuits_one_launch_full_width(ctx):
    """The many-circuits configuration at its real width (1 024 vectors per circuit, circuits of
    64 to 4 096 constraints, hundreds of tapes in one launch): nothing fails, the per-tape counts
    of planted failures come back on the right tapes, and sampled tapes match the checker."""
    torch = pytest.importorskip("torch")
    from zkir_amd import campaign
    n_circ, n_vec = 384, 1024
    tapes = []
    for cid in range(n_circ):
        tapes.append(z.Lowerer(z.Generator(campaign.circuit_seed(77, cid), campaign.campaign_config(cid)).generate()).lower())
    assert min(t.hdr.n_constraints for t in tapes) < 100 and max(t.hdr.n_constraints for t in tapes) > 3000
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    words = sum(t.hdr.n_inputs for t in tapes) * 8 * n_vec
    inp = torch.randint(0, 1 << 32, (words,), dtype=torch.int64, device=dev, generator=gen).to(torch.int32)
    offs, off = [], 0
    for t in tapes:
        blk = inp[off:off + t.hdr.n_inputs * 8 * n_vec].view(t.hdr.n_inputs, 8, n_vec)
        blk[:, 7, :] &= 0x0FFFFFFF
        offs.append(off)
        off += t.hdr.n_inputs * 8 * n_vec
    # plant failures: in three tapes turn the first product into a sum
    planted = {5: None, 200: None, 383: None}
    for k in planted:
        ops = tapes[k].ops()
        j = next(i for i, op in enumerate(ops) if op[0] == F.OP_MUL)
        ops[j] = (F.OP_ADD,) + ops[j][1:]
        blob = b"".join(struct.pack("<4I", *op) for op in ops) + tapes[k].blob()[16 * len(ops):]
        tapes[k] = z.Tape.from_parts(tapes[k].hdr, blob)
    tids = [ctx.upload(t) for t in tapes]
    unsat = torch.zeros(n_circ, dtype=torch.int64, device=dev)
    ctx.eval_multi_dev(tids, inp.data_ptr(), n_vec, unsat.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = [int(v) for v in unsat.cpu()]
    for k in list(planted) + [0, 100, 382]:  # the planted tapes and three clean ones, against the checker
        t = tapes[k]
        blk = inp[offs[k]:offs[k] + t.hdr.n_inputs * 8 * n_vec].view(t.hdr.n_inputs, 8, n_vec)
        blk = np.ascontiguousarray(blk.cpu().numpy().view(np.uint32))
        _, u_c, _, _ = C.eval_cpu(bytes(t.hdr), t.blob(), blk, n_vec, t.hdr.n_outputs, want_outputs=False)
        assert got[k] == u_c, k
    assert all(got[k] > 0 for k in planted)
    assert all(g == 0 for k, g in enumerate(got) if k not in planted)
    for t in tids:
        ctx.release(t)


@pytest.mark.parametrize("strands", [0, 4, 8])
def test_deep_gadget_full_size(ctx, strands, interpreter_kernel):
    """The deep chain at its real size: 250 008 constraints over 2^16 vectors, every constraint
    of every vector holds, and the last outputs of sampled vectors match the checker; as a plain
    tape and as a strand tape (four wavefronts per 64 vectors, the shape for a batch this small)."""
    torch = pytest.importorskip("torch")
    if strands and interpreter_kernel == "compiled":
        pytest.skip("strand tapes run on the generated stream only")
    o = z.default_lower_options()
    o.strands = strands
    tape = z.Lowerer(z.poseidon_style_gadget(4, 20834, 7), o).lower()
    h = tape.hdr
    assert h.n_constraints == 250008 and bool(h.flags & 1) == bool(strands)
    n = 1 << 16
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(11)
    inp = torch.randint(0, 1 << 32, (h.n_inputs, 8, n), dtype=torch.int64, device=dev, generator=gen).to(torch.int32)
    inp[:, 7, :] &= 0x0FFFFFFF
    res = torch.zeros(4, dtype=torch.int64, device=dev)
    tid = ctx.upload(tape)
    ctx.eval_batch_dev(tid, inp.data_ptr(), n, None, res.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    r = res.cpu().numpy().view(np.uint64)
    assert r[0] == 0 and r[1] == np.uint64(0xFFFFFFFFFFFFFFFF) and r[2] == 0
    # a handful of vectors through the checker, outputs included (the chain's end depends on every round)
    idx = torch.tensor([0, 1, 255, 256, 32767, 65535], device=dev)
    inp_s = np.ascontiguousarray(inp[:, :, idx].cpu().numpy().view(np.uint32))
    m = len(idx)
    out = torch.empty((h.n_outputs, 8, m), dtype=torch.int32, device=dev)
    inp_d = torch.from_numpy(inp_s.view(np.int32)).to(dev)
    ctx.eval_batch_dev(tid, inp_d.data_ptr(), m, out.data_ptr(), res.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    o_c, u_c, _, _ = C.eval_cpu(bytes(h), tape.blob(), inp_s, m, h.n_outputs, threads=6)
    assert u_c == 0 and np.array_equal(np.ascontiguousarray(out.cpu().numpy().view(np.uint32)), o_c)
    ctx.release(tid)


def test_batch_beyond_the_generated_stream_limit(ctx):
    """32*N no longer fits 32 bits: the launch takes the compiled kernel (64-bit offsets), same results."""
    torch = pytest.importorskip("torch")
    ops = [(F.OP_INPUT, G(0), I(0), NONE), (F.OP_MUL, G(1), G(0), K(0)), (F.OP_CHECK, G(1), G(0), K(0)),
           (F.OP_CHECK, G(0), G(0), K(1))]  # x*3 == x*3 holds; x*1 == x holds
    tape = hand_tape(ops, [3, 1], 1, 0, 2, 0)
    n = (1 << 27) + 5
    dev = torch.device("cuda:0")
    inp = torch.empty((1, 8, n), dtype=torch.int32, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(2)
    for l in range(8):  # limb by limb: a full-size int64 temporary would be 8 GiB
        inp[0, l] = torch.randint(0, 1 << 32, (n,), dtype=torch.int64, device=dev, generator=gen).to(torch.int32)
    inp[0, 7] &= 0x0FFFFFFF
    res = torch.zeros(4, dtype=torch.int64, device=dev)
    tid = ctx.upload(tape)
    ctx.eval_batch_dev(tid, inp.data_ptr(), n, None, res.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    r = res.cpu().numpy().view(np.uint64)
    assert r[0] == 0 and r[1] == np.uint64(0xFFFFFFFFFFFFFFFF)
    # the same tape with the wrong constant in its check fails for every vector whose input is not 0
    bad = hand_tape([(F.OP_INPUT, G(0), I(0), NONE), (F.OP_MUL, G(1), G(0), K(0)), (F.OP_CHECK, G(1), G(0), K(1))],
                    [3, 1], 1, 0, 2, 0)
    tid2 = ctx.upload(bad)
    ctx.eval_batch_dev(tid2, inp.data_ptr(), n, None, res.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    r = res.cpu().numpy().view(np.uint64)
    zeros = int((inp[0] == 0).all(dim=0).sum())
    assert int(r[0]) == n - zeros
    ctx.release(tid)
    ctx.release(tid2)
    del inp
    torch.cuda.empty_cache()


def test_measured_rates_are_sane(ctx):
    mulhi = ctx.int_mul_rate(0)
    assert 5e12 < mulhi < 6e13
    fr = ctx.fr_mul_rate()
    assert 2e10 < fr < 4e11


def test_two_streams_one_context_give_serial_results(ctx):
    """Launches of one context share its work counters, descriptors and slot pool; the library chains
    them, so two device-pointer calls issued back to back on two streams give what serial calls give."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    cfg = dict(forced_inputs=6, inputs_readable=1, arithmetic_only=1, quadratic_aware=1, no_loops=1, safe_array_index=1)
    ta = z.Lowerer(z.Generator(501, ext_config(min_instructions=2500, max_instructions=2500, **cfg)).generate()).lower()
    tb = z.Lowerer(z.Generator(502, ext_config(min_instructions=900, max_instructions=900, **cfg)).generate()).lower()
    # tb with a failing constraint planted, so the counts are not all zero
    ops = tb.ops()
    j = next(i for i, op in enumerate(ops) if op[0] == F.OP_MUL)
    ops[j] = (F.OP_ADD,) + ops[j][1:]
    tb = z.Tape.from_parts(tb.hdr, b"".join(struct.pack("<4I", *op) for op in ops) + tb.blob()[16 * len(ops):])
    ida, idb = ctx.upload(ta), ctx.upload(tb)
    na, nb = 150000, 40000
    gen = torch.Generator(device=dev)
    gen.manual_seed(21)

    def inputs(t, n):
        a = torch.randint(0, 1 << 32, (t.hdr.n_inputs, 8, n), dtype=torch.int64, device=dev, generator=gen).to(torch.int32)
        a[:, 7, :] &= 0x0FFFFFFF
        return a

    ina, inb = inputs(ta, na), inputs(tb, nb)
    many = [z.Lowerer(z.Generator(600 + k, ext_config(min_instructions=300, max_instructions=300, **cfg)).generate()).lower()
            for k in range(40)]
    mids = ctx.upload_many(many)
    nm = 512
    inm = torch.randint(0, 1 << 32, (sum(t.hdr.n_inputs for t in many), 8, nm), dtype=torch.int64, device=dev,
                        generator=gen).to(torch.int32)
    inm[:, 7, :] &= 0x0FFFFFFF

    def run(streams):
        oa = torch.zeros((ta.hdr.n_outputs, 8, na), dtype=torch.int32, device=dev)
        ob = torch.zeros((tb.hdr.n_outputs, 8, nb), dtype=torch.int32, device=dev)
        ra, rb, rm = (torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(3))
        um = torch.zeros(len(many), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        ctx.eval_batch_dev(ida, ina.data_ptr(), na, oa.data_ptr(), ra.data_ptr(), streams[0].cuda_stream)
        ctx.eval_batch_dev(idb, inb.data_ptr(), nb, ob.data_ptr(), rb.data_ptr(), streams[1].cuda_stream)
        ctx.eval_multi_dev(mids, inm.data_ptr(), nm, um.data_ptr(), streams[2].cuda_stream, rm.data_ptr())
        ctx.eval_batch_dev(ida, ina.data_ptr(), na, oa.data_ptr(), ra.data_ptr(), streams[1].cuda_stream)
        torch.cuda.synchronize()
        return oa.cpu(), ob.cpu(), ra.cpu(), rb.cpu(), um.cpu(), rm.cpu()

    s0 = torch.cuda.current_stream()
    serial = run([s0, s0, s0])
    side = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
    overlapped = run(side)
    for a, b in zip(serial, overlapped):
        assert torch.equal(a, b)
    assert int(serial[2][0]) == 0 and int(serial[3][0]) > 0 and int(serial[4].sum()) == 0
    # sampled vectors of the first call against the checker
    idx = [0, 1, 77777, na - 1]
    inp_s = np.ascontiguousarray(ina[:, :, idx].cpu().numpy().view(np.uint32))
    o_c, _, _, _ = C.eval_cpu(bytes(ta.hdr), ta.blob(), inp_s, len(idx), ta.hdr.n_outputs)
    assert np.array_equal(np.ascontiguousarray(overlapped[0][:, :, idx].numpy().view(np.uint32)), o_c)
    for t in [ida, idb] + mids:
        ctx.release(t)


def test_release_while_later_launches_run(ctx):
    """A pipelined caller releases the tapes of a launch while the next launches are still running (on a stream of its
    own) and uploads new tapes into the space: the release waits for the launch that read the tape, not for the
    device, and nothing a running launch reads is touched.  Results equal those of the same work done serially."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    cfg = dict(forced_inputs=6, inputs_readable=1, arithmetic_only=1, quadratic_aware=1, no_loops=1, safe_array_index=1)
    groups = [[z.Lowerer(z.Generator(700 + 10 * g + k, ext_config(min_instructions=400 + 100 * k, max_instructions=400 + 100 * k, **cfg)).generate()).lower()
               for k in range(6)] for g in range(12)]  # more groups than the runtime keeps launch events for
    n = 4096
    gen = torch.Generator(device=dev)
    gen.manual_seed(33)
    inputs = []
    for tapes in groups:
        a = torch.randint(0, 1 << 32, (sum(t.hdr.n_inputs for t in tapes), 8, n), dtype=torch.int64, device=dev, generator=gen).to(torch.int32)
        a[:, 7, :] &= 0x0FFFFFFF
        inputs.append(a)
    # plant a failure in one tape of every group, so that the counts differ from group to group
    for g, tapes in enumerate(groups):
        ops = [list(op) for op in tapes[g % 6].ops()]
        j = [i for i, op in enumerate(ops) if op[0] & 0xFF == F.OP_CHECK][g]
        ops[j][1] = F.operand(F.K_CONST, 0)  # the product is compared with the pool's first constant
        tapes[g % 6] = z.Tape.from_parts(tapes[g % 6].hdr, b"".join(struct.pack("<4I", *op) for op in ops) + tapes[g % 6].blob()[16 * len(ops):])

    def serial():
        out = []
        for tapes, inp in zip(groups, inputs):
            ids = ctx.upload_many(tapes)
            u = torch.zeros(len(tapes), dtype=torch.int64, device=dev)
            ctx.eval_multi_dev(ids, inp.data_ptr(), n, u.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            out.append(u.cpu())
            for i in ids:
                ctx.release(i)
        return out

    def pipelined():
        side = torch.cuda.Stream()
        out, prev = [], None
        for tapes, inp in zip(groups, inputs):
            ids = ctx.upload_many(tapes)  # while the last group's launch runs
            u = torch.zeros(len(tapes), dtype=torch.int64, device=dev)
            side.wait_stream(torch.cuda.current_stream())
            ctx.eval_multi_dev(ids, inp.data_ptr(), n, u.data_ptr(), side.cuda_stream)
            done = torch.cuda.Event()
            done.record(side)
            if prev is not None:  # the last group: wait for ITS launch only, read, release while this group's launch runs
                pids, pu, pdone = prev
                pdone.synchronize()
                out.append(pu.cpu())
                for i in pids:
                    ctx.release(i)
            prev = (ids, u, done)
        pids, pu, pdone = prev
        pdone.synchronize()
        out.append(pu.cpu())
        for i in pids:
            ctx.release(i)
        return out

    want = serial()
    assert all(int(u.sum()) > 0 for u in want)
    for _ in range(2):
        # a tape launched before all of them and released after: its launch is older than the events the runtime keeps
        first = ctx.upload(groups[0][0])
        u0 = torch.zeros(1, dtype=torch.int64, device=dev)
        ctx.eval_multi_dev([first], inputs[0].data_ptr(), n, u0.data_ptr(), torch.cuda.current_stream().cuda_stream)
        got = pipelined()
        ctx.release(first)
        assert len(got) == len(want) and all(torch.equal(a, b) for a, b in zip(got, want))
        assert int(u0[0]) == int(want[0][0])


def test_run_time_faults_of_a_multi_tape_launch_reach_the_caller(ctx):
    """Division by a run-time zero in some vectors of some tapes of one device-pointer launch: the fault
    count comes back in the caller's result words (and through the campaign's evaluator)."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    div = hand_tape([(F.OP_INPUT, G(0), I(0), NONE), (F.OP_INPUT, G(1), I(1), NONE), (F.OP_DIV, G(2), G(0), G(1)),
                     (F.OP_OUTPUT, 0, G(2), NONE)], [], 2, 1, 3, 0)
    rem = hand_tape([(F.OP_INPUT, G(0), I(0), NONE), (F.OP_INPUT, G(1), I(1), NONE), (F.OP_REM, G(2), G(0), G(1)),
                     (F.OP_MUL, G(3), G(2), G(0)), (F.OP_CHECK, G(3), G(2), G(0))], [], 2, 0, 4, 0)
    clean = z.Lowerer(z.Generator(9, ext_config(min_instructions=120, max_instructions=120, forced_inputs=2, inputs_readable=1,
                                                 arithmetic_only=1, quadratic_aware=1, no_loops=1, safe_array_index=1)).generate()).lower()
    tapes = [clean, div, clean, rem, div]
    n = 700
    rnd = random.Random(31)
    blocks, want_faults, want_per_tape = [], 0, []
    for j, t in enumerate(tapes):
        vecs = random_vectors(rnd, t.hdr.n_inputs, n, special_share=0.0)
        if t is not clean:
            for k in rnd.sample(range(n), 7 + j):  # a different count on every faulting tape
                vecs[k][1] = 0
        inp = z.pack_inputs(vecs, t.hdr.n_inputs)
        _, _, _, f_c = C.eval_cpu(bytes(t.hdr), t.blob(), inp, n, t.hdr.n_outputs, want_outputs=False)
        want_faults += f_c
        want_per_tape.append(f_c)
        blocks.append(inp.reshape(-1))
    assert want_per_tape == [0, 8, 0, 10, 11]
    tids = ctx.upload_many(tapes)
    inp = torch.from_numpy(np.concatenate(blocks).view(np.int32)).to(dev)
    unsat = torch.zeros(len(tapes), dtype=torch.int64, device=dev)
    per_tape = torch.full((len(tapes),), 3, dtype=torch.int64, device=dev)
    res = torch.full((4,), 5, dtype=torch.int64, device=dev)
    ctx.eval_multi_dev(tids, inp.data_ptr(), n, unsat.data_ptr(), torch.cuda.current_stream().cuda_stream, res.data_ptr())
    torch.cuda.synchronize()
    r = res.cpu().numpy().view(np.uint64)
    assert int(r[2]) == want_faults and int(r[0]) == 0 and int(unsat.sum()) == 0
    # the same launch with a counter per tape: the checker's count on every tape
    ctx.eval_multi_dev(tids, inp.data_ptr(), n, unsat.data_ptr(), torch.cuda.current_stream().cuda_stream, res.data_ptr(),
                       per_tape.data_ptr())
    torch.cuda.synchronize()
    assert [int(v) for v in per_tape.cpu()] == want_per_tape and int(res.cpu().numpy().view(np.uint64)[2]) == want_faults
    # the host-pointer form reports the same count
    _, faults = ctx.eval_multi(tids, np.concatenate(blocks), n)
    assert faults == want_faults
    for t in tids:
        ctx.release(t)
    # and the campaign's evaluator no longer drops them: draws are random, so plant a zero divisor as a constant
    from zkir_amd import campaign
    ev = campaign.GpuEvaluator(0, input_seed=3)
    zero_div = hand_tape([(F.OP_INPUT, G(0), I(0), NONE), (F.OP_IDIV, G(1), G(0), K(0)), (F.OP_OUTPUT, 0, G(1), NONE)],
                         [0], 1, 1, 2, 0)
    got_unsat, got_faults = ev([clean, zero_div, clean], 300, [0, 1, 2])
    assert got_unsat == [0, 0, 0] and got_faults == [0, 300, 0]
    # run_local flags the circuit that faulted, not the first one of its launch
    res_c = campaign.run_local([10, 11, 12], 1, 300, lambda tapes, nv, ids: ev([clean, zero_div, clean], nv, ids),
                               lowered=iter([(10, 0, clean), (11, 0, zero_div), (12, 0, clean)]))
    assert res_c.divergent == [11] and res_c.faults == 300 and res_c.unsat == 0
    ev.close()


def test_tape_with_more_device_slots_than_the_generated_stream_addresses(ctx):
    """8 KiB x n_gslots no longer fits the 32-bit offsets of the generated stream: the launch takes the
    compiled kernel (64-bit addresses) and the far slots hold what was put there."""
    torch = pytest.importorskip("torch")
    far = (1 << 19) + 2  # slot rows of 8 or 9 KiB: the last one sits past 4 GiB in the workgroup's region
    ops = [(F.OP_INPUT, G(far), I(0), NONE), (F.OP_INPUT, G(1), I(1), NONE), (F.OP_MUL, G(far - 1), G(far), G(1)),
           (F.OP_ADD, G(2), G(far - 1), G(far)), (F.OP_OUTPUT, 0, G(2), NONE), (F.OP_OUTPUT, 1, G(far), NONE),
           (F.OP_CHECK, G(far - 1), G(1), G(far))]
    tape = hand_tape(ops, [], 2, 2, far + 1, 0)
    vecs = random_vectors(random.Random(8), 2, 200)
    out, unsat, _, _ = check_against_c_oracle(ctx, tape, vecs)
    assert unsat == 0 and z.unpack_outputs(out[:, :, 5:6])[0] == [F.add(F.mul(vecs[5][0], vecs[5][1]), vecs[5][0]), vecs[5][0]]
    torch.cuda.empty_cache()


def test_circuits_with_if_statements_against_both_checkers(ctx):
    """The generator's If statements with conditions lower can decide (enable_if = 2): the tape of the taken
    branches on the GPU against the C checker, and against the direct interpretation of the AST."""
    n_ok = n_if = 0
    for seed in range(1, 60):
        c = z.Generator(seed, ext_config(min_instructions=120, max_instructions=160, forced_inputs=4, inputs_readable=1,
                                         safe_array_index=1, arithmetic_only=seed % 2, quadratic_aware=1, no_loops=seed % 3 == 0,
                                         enable_if=2)).generate()
        txt = c.text()
        o = z.default_lower_options()
        o.max_iterations = 300
        try:
            tape = z.Lowerer(c, o).lower()
        except z.LowerRejected:
            continue
        n_ok += 1
        n_if += "(if " in txt
        vecs = random_vectors(random.Random(seed), tape.hdr.n_inputs, 64)
        out, unsat, first, faults = check_against_c_oracle(ctx, tape, vecs)
        got = z.unpack_outputs(out)
        for t in (0, 63):
            it = R.Interpreter(R.parse_text(txt), vecs[t], max_iterations=300).run()
            assert got[t] == it.outputs and it.unsat() == []
        assert unsat == 0
    assert n_ok >= 10 and n_if >= 5


def test_campaign_sized_launch_with_planted_failures_and_faults(ctx, interpreter_kernel):
    """2 304 distinct campaign circuits x 1 024 vectors in ONE device-pointer launch (a campaign's launch size), two
    every-operator tapes with a zero divisor among them: per-tape counts on the right tapes, the fault count in the
    caller's result words, sampled tapes against the C checker."""
    torch = pytest.importorskip("torch")
    if interpreter_kernel != "generated-stream":
        pytest.skip("one pass at this size is enough; the 384-tape test runs every kernel")
    from zkir_amd import campaign
    n_circ, n_vec = 2304, 1024
    tapes = [t for _, cls, t in campaign.lower_parallel(range(n_circ), 4242, workers=8) if cls == 0]
    assert len(tapes) == n_circ
    planted = [3, 777, 1500, 2303]
    for k in planted:  # turn the first product into a sum: the circuit's first constraint fails
        ops = tapes[k].ops()
        j = next(i for i, op in enumerate(ops) if op[0] == F.OP_MUL)
        ops[j] = (F.OP_ADD,) + ops[j][1:]
        tapes[k] = z.Tape.from_parts(tapes[k].hdr, b"".join(struct.pack("<4I", *op) for op in ops) + tapes[k].blob()[16 * len(ops):])
    zero_div = hand_tape([(F.OP_INPUT, G(0), I(0), NONE), (F.OP_REM, G(1), G(0), K(0)), (F.OP_OUTPUT, 0, G(1), NONE)], [0], 1, 1, 2, 0)
    fault_at = [n_circ, n_circ + 1]  # appended: a launch that holds them takes the every-operator kernel for all its tapes
    tapes += [zero_div, zero_div]
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(9)
    words = sum(t.hdr.n_inputs for t in tapes) * 8 * n_vec
    inp = torch.randint(0, 1 << 32, (words,), dtype=torch.int64, device=dev, generator=gen).to(torch.int32)
    inp.view(-1, 8, n_vec)[:, 7, :] &= 0x0FFFFFFF
    offs, off = [], 0
    for t in tapes:
        offs.append(off)
        off += t.hdr.n_inputs * 8 * n_vec
    tids = ctx.upload_many(tapes)
    unsat = torch.zeros(len(tapes), dtype=torch.int64, device=dev)
    res = torch.zeros(4, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    # the + - * tapes alone (the generated stream), then all of them (the compiled every-operator kernel)
    ctx.eval_multi_dev(tids[:n_circ], inp.data_ptr(), n_vec, unsat.data_ptr(), stream, res.data_ptr())
    torch.cuda.synchronize()
    got = [int(v) for v in unsat.cpu()][:n_circ]
    assert int(res.cpu().numpy().view(np.uint64)[2]) == 0
    ctx.eval_multi_dev(tids, inp.data_ptr(), n_vec, unsat.data_ptr(), stream, res.data_ptr())
    torch.cuda.synchronize()
    assert [int(v) for v in unsat.cpu()] == got + [0, 0]
    assert int(res.cpu().numpy().view(np.uint64)[2]) == len(fault_at) * n_vec
    for k in planted + [0, 1, 1151, 2302]:
        t = tapes[k]
        blk = inp[offs[k]:offs[k] + t.hdr.n_inputs * 8 * n_vec].view(t.hdr.n_inputs, 8, n_vec)
        blk = np.ascontiguousarray(blk.cpu().numpy().view(np.uint32))
        _, u_c, _, _ = C.eval_cpu(bytes(t.hdr), t.blob(), blk, n_vec, t.hdr.n_outputs, want_outputs=False, threads=8)
        assert got[k] == u_c, k
    # (a changed product need not reach a constraint in every circuit: the checker's count above is the reference)
    assert sum(1 for k in planted if got[k] > 0) >= 1 and all(g == 0 for k, g in enumerate(got) if k not in planted)
    for t in tids:
        ctx.release(t)


@pytest.mark.gpu
def test_reducer_over_rccl_sums_counters_and_unites_ids():
    """zkir_reducer_*: the campaign's exchange step behind the C ABI (RCCL communicators made in this process).
    With the one GPU of this box the collectives run over a single rank; the sum, the count-first padded gather,
    the sorted union and the capacity check are the same code for any number of devices."""
    from zkir_amd import api
    r = api.Reducer([0], 6, 16)
    try:
        totals, ids = r.reduce([[5, 0, 2**40 + 3, 1, 0, 7]], [[9, 3, 3, 12]])
        assert totals == [5, 0, 2**40 + 3, 1, 0, 7] and ids == [3, 9, 12]
        totals, ids = r.reduce([[1, 2, 3, 4, 5, 6]], [[]])          # a second call, nothing divergent
        assert totals == [1, 2, 3, 4, 5, 6] and ids == []
        with pytest.raises(api.ZkirError) as e:
            r.reduce([[0] * 6], [list(range(17))])
        assert e.value.code == api.E_LIMIT
    finally:
        r.close()
