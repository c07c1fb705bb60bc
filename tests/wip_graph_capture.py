"""zkir_eval_batch_dev_noalloc recorded into a HIP graph and replayed, and the refusal of a second context's
call while the first one's launch still runs.  Needs a real MI355X."""
import numpy as np
import pytest

import zkir_amd as z
from zkir_amd import api

pytestmark = pytest.mark.gpu


def _cfg(n_instr, inputs):
    g = z.default_generator_config()
    g.min_instructions = g.max_instructions = n_instr
    g.forced_inputs, g.inputs_readable, g.arithmetic_only, g.quadratic_aware = inputs, 1, 1, 1
    g.no_loops, g.safe_array_index = 1, 1
    return g


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
