"""Where the headline launch's time goes, by taking costs away from its tape one at a time (results of the altered
tapes are wrong by construction; only their times mean something).  Run on a GPU box:
python tools/probes/headline_breakdown.py"""
import os, sys, struct, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import zkir_amd as z, bench
from zkir_amd import campaign

K = lambda w: int(w) >> 29
NONE, L0 = 5 << 29, 1 << 29


def rebuild(t, ops):
    hdr = z.api.TapeHdr.from_buffer_copy(bytes(t.hdr))
    hdr.n_ops = len(ops)
    return z.Tape.from_parts(hdr, b"".join(struct.pack("<4I", *op) for op in ops) + t.blob()[16 * t.hdr.n_ops:])


def no_spill(t):
    """every device-slot read comes from window row 0, every device-slot write is dropped"""
    ops = []
    for code, dst, a, b in t.ops():
        c = code & 0xff
        if c == 3 and K(dst) == 0 and K(a) == 4:
            continue  # COPY of the accumulator to a device slot
        if c not in (2, 9, 48) and K(dst) == 0:
            dst = NONE
        if c == 48 and K(dst) == 0:
            dst = L0
        if c not in (1, 8) and K(a) == 0:
            a = L0
        if K(b) == 0:
            b = L0
        ops.append((code, dst, a, b))
    return rebuild(t, ops)


def plain_stores(t):
    """every OUTPUT a store without the conversion"""
    return rebuild(t, [((code & ~0xff) | 9 if code & 0xff == 2 else code, dst, a, b) for code, dst, a, b in t.ops()])


def timeit(ctx, tape, label, want_outputs=True, n=1 << 20, steps=5):
    dev = torch.device("cuda:0")
    h = tape.hdr
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    inp = campaign.uniform_inputs(torch, h.n_inputs, n, dev, gen)
    out = torch.empty((h.n_outputs, 8, n), dtype=torch.int32, device=dev) if want_outputs else None
    res = torch.zeros(4, dtype=torch.int64, device=dev)
    tid = ctx.upload(tape)
    s = torch.cuda.current_stream().cuda_stream
    op = out.data_ptr() if out is not None else None
    for _ in range(2):
        ctx.eval_batch_dev(tid, inp.data_ptr(), n, op, res.data_ptr(), s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.eval_batch_dev(tid, inp.data_ptr(), n, op, res.data_ptr(), s)
    torch.cuda.synchronize()
    print("%-64s %7.2f ms  (%d ops, %d device slots)" % (label, (time.perf_counter() - t0) / steps * 1e3, h.n_ops, h.n_gslots), flush=True)
    ctx.release(tid)


os.environ["ZKIR_INTERP"] = os.environ.get("ZKIR_BREAKDOWN_KERNEL", "lean32")  # the altered tapes keep few device slots: hold the kernel fixed
ctx = z.Context(0)
t = bench.workload_tape(z)
for rep in range(2):
    timeit(ctx, t, "the headline tape")
    timeit(ctx, no_spill(t), "no spill (slot reads from one window row, slot writes dropped)")
    timeit(ctx, plain_stores(t), "outputs stored without conversion")
    timeit(ctx, plain_stores(no_spill(t)), "no spill, outputs stored without conversion")
    timeit(ctx, no_spill(t), "no spill, outputs not requested", want_outputs=False)
