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


def _inputs(torch, n_inputs, n, seed):
    from zkir_amd import campaign
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    return campaign.uniform_inputs(torch, n_inputs, n, dev, gen)


def test_no_alloc_entry_point_recorded_into_a_graph_and_replayed():
    torch = pytest.importorskip("torch")
    ctx = z.Context(0)
    try:
        tape = z.Lowerer(z.Generator(4242, _cfg(1500, 6)).generate()).lower()
        h = tape.hdr
        assert 0 < h.n_gslots <= api.lib_noalloc_slot_rows()
        tid = ctx.upload(tape)
        n = 20000
        dev = torch.device("cuda:0")
        batches = [_inputs(torch, h.n_inputs, n, seed) for seed in (11, 12)]
        want = []
        for b in batches:  # what the allocating entry point gives for each batch
            out = torch.zeros((h.n_outputs, 8, n), dtype=torch.int32, device=dev)
            res = torch.zeros(4, dtype=torch.int64, device=dev)
            ctx.eval_batch_dev(tid, b.data_ptr(), n, out.data_ptr(), res.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            want.append((out.cpu().numpy().copy(), res.cpu().numpy().copy()))
        inp = batches[0].clone()
        out = torch.full((h.n_outputs, 8, n), 0x5A5A5A5A, dtype=torch.int32, device=dev)
        res = torch.full((4,), -7, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()  # the upload and every earlier launch are done before the capture begins
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            ctx.eval_batch_dev_noalloc(tid, inp.data_ptr(), n, out.data_ptr(), res.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert int(res[0].item()) == -7, "recording a graph runs nothing"
        for k in (0, 1, 0):  # replays on fresh inputs written into the recorded buffers
            inp.copy_(batches[k])
            out.fill_(0x5A5A5A5A)
            res.fill_(-7)
            graph.replay()
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), want[k][0]), k
            assert np.array_equal(res.cpu().numpy(), want[k][1]), k
        del graph
        ctx.release(tid)
    finally:
        ctx.close()


def test_second_context_is_refused_while_the_first_one_runs_on_the_scratch_array():
    torch = pytest.importorskip("torch")
    a, b = z.Context(0), z.Context(0)
    try:
        so = z.default_lower_options()
        so.strands = 0
        long_tape = z.Lowerer(z.poseidon_style_gadget(4, 3000, 3), so).lower()  # tens of milliseconds per launch
        ta, tb = a.upload(long_tape), b.upload(long_tape)
        n = 1 << 16
        inp = _inputs(torch, long_tape.hdr.n_inputs, n, 5)
        res_a = torch.zeros(4, dtype=torch.int64, device=inp.device)
        res_b = torch.zeros(4, dtype=torch.int64, device=inp.device)
        torch.cuda.synchronize()
        s = torch.cuda.current_stream().cuda_stream
        a.eval_batch_dev_noalloc(ta, inp.data_ptr(), n, None, res_a.data_ptr(), s)
        rc = api.lib.zkir_eval_batch_dev_noalloc(b._h, tb, inp.data_ptr(), n, None, res_b.data_ptr(), s)
        assert rc == api.E_BUSY and b"still running" in api.lib.zkir_last_error(b._h)
        a.eval_batch_dev_noalloc(ta, inp.data_ptr(), n, None, res_a.data_ptr(), s)  # the same context is chained, not refused
        torch.cuda.synchronize()
        b.eval_batch_dev_noalloc(tb, inp.data_ptr(), n, None, res_b.data_ptr(), s)  # the array is free again
        torch.cuda.synchronize()
        assert int(res_a[0].item()) == 0 and int(res_b[0].item()) == 0
        a.release(ta)
        b.release(tb)
    finally:
        a.close()
        b.close()
