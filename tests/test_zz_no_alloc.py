"""zkir_eval_batch_dev_noalloc (the entry point for callers that cannot allocate: work counter and
witness slots in the library's fixed scratch array) against zkir_eval_batch_dev, through the C ABI,
bit for bit.  Needs a real MI355X."""
import ctypes
import random

import numpy as np
import pytest

import zkir_amd as z
from oracle import fr_ref as F
from zkir_amd import api

pytestmark = pytest.mark.gpu


def _cfg(n_instr, inputs):
    g = z.default_generator_config()
    g.min_instructions = g.max_instructions = n_instr
    g.forced_inputs, g.inputs_readable, g.arithmetic_only, g.quadratic_aware = inputs, 1, 1, 1
    g.no_loops, g.safe_array_index = 1, 1
    return g


def _device_inputs(torch, n_inputs, n, seed):
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    inp = torch.randint(0, 1 << 32, (max(n_inputs, 1), 8, n), dtype=torch.int64, device=dev, generator=gen).to(torch.int32)
    inp[:, 7, :] &= 0x0FFFFFFF  # every element below 2^252 < p
    return inp


def _both(torch, ctx, tid, hdr, inp, n):
    """(outputs, result words) of the allocating entry point and of the no-alloc one on the same inputs."""
    dev = inp.device
    got = []
    stream = torch.cuda.current_stream().cuda_stream
    for fn in (ctx.eval_batch_dev, ctx.eval_batch_dev_noalloc):
        out = torch.full((hdr.n_outputs, 8, n), 0x5A5A5A5A, dtype=torch.int32, device=dev)
        res = torch.full((4,), -7, dtype=torch.int64, device=dev)
        fn(tid, inp.data_ptr(), n, out.data_ptr(), res.data_ptr(), stream)
        torch.cuda.synchronize()
        got.append((out.cpu().numpy().view(np.uint32), res.cpu().numpy().view(np.uint64)))
    return got


def test_no_alloc_entry_point_matches_the_allocating_one():
    torch = pytest.importorskip("torch")
    ctx = z.Context(0)
    try:
        # a circuit with spilled witness slots (device-memory rows) and failing constraints planted in it
        tape = z.Lowerer(z.Generator(4242, _cfg(1500, 6)).generate()).lower()
        h = tape.hdr
        assert 0 < h.n_gslots <= api.lib_noalloc_slot_rows()
        ops = tape.ops()
        j = [i for i, op in enumerate(ops) if op[0] == F.OP_MUL][3]
        ops[j] = (F.OP_ADD,) + ops[j][1:]
        import struct
        blob = b"".join(struct.pack("<4I", *op) for op in ops) + tape.blob()[16 * len(ops):]
        broken = z.Tape.from_parts(h, blob)
        tid, tid_broken = ctx.upload(tape), ctx.upload(broken)
        # two calls in a row on the shared scratch, the second with a smaller input (and the other tape)
        for which, n, seed in ((tid, 70000, 1), (tid_broken, 333, 2), (tid, 1, 3)):
            inp = _device_inputs(torch, h.n_inputs, n, seed)
            (o_a, r_a), (o_b, r_b) = _both(torch, ctx, which, h, inp, n)
            assert np.array_equal(o_a, o_b), (which, n)
            assert np.array_equal(r_a, r_b), (which, n, r_a, r_b)
            if which == tid:
                assert r_a[0] == 0 and r_a[1] == np.uint64(0xFFFFFFFFFFFFFFFF)
            else:
                assert r_a[0] > 0
        # a strand tape takes the same path (four wavefronts per 64 vectors, per-strand slot rows)
        so = z.default_lower_options()
        so.strands = 4
        gadget = z.Lowerer(z.poseidon_style_gadget(4, 40, 3), so).lower()
        assert gadget.hdr.flags & 1
        gid = ctx.upload(gadget)
        for n, seed in ((5000, 4), (65, 5)):
            inp = _device_inputs(torch, gadget.hdr.n_inputs, n, seed)
            (o_a, r_a), (o_b, r_b) = _both(torch, ctx, gid, gadget.hdr, inp, n)
            assert np.array_equal(o_a, o_b) and np.array_equal(r_a, r_b) and r_a[0] == 0
        # beyond the fixed scratch: an error code and a message, not a launch
        rows = api.lib_noalloc_slot_rows()
        G, I, NONE = (lambda i: F.operand(F.K_GSLOT, i)), (lambda i: F.operand(F.K_INPUT, i)), F.operand(F.K_NONE)
        big_ops = [(F.OP_INPUT, G(rows), I(0), NONE), (F.OP_OUTPUT, 0, G(rows), NONE)]
        big_hdr = api.TapeHdr(0x31544B5A, 2, 1, 1, 0, len(big_ops), rows + 1, 0, 0, 0, 0)
        big = z.Tape.from_parts(big_hdr, b"".join(struct.pack("<4I", *o) for o in big_ops))
        bid = ctx.upload(big)
        inp = _device_inputs(torch, 1, 10, 6)
        out = torch.zeros((1, 8, 10), dtype=torch.int32, device=inp.device)
        res = torch.zeros(4, dtype=torch.int64, device=inp.device)
        rc = api.lib.zkir_eval_batch_dev_noalloc(ctx._h, bid, inp.data_ptr(), 10, out.data_ptr(), res.data_ptr(), None)
        assert rc == api.E_LIMIT and b"fixed scratch" in api.lib.zkir_last_error(ctx._h)
        # the allocating entry point takes it
        ctx.eval_batch_dev(bid, inp.data_ptr(), 10, out.data_ptr(), res.data_ptr(), None)
        torch.cuda.synchronize()
        assert torch.equal(out, inp)
        for t in (tid, tid_broken, gid, bid):
            ctx.release(t)
    finally:
        ctx.close()
