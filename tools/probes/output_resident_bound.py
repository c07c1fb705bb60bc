"""Run on a GPU box: [ZKIR_SCALE_CONVERT_COST=0.1] python tools/probes/output_resident_bound.py plain|scaled
Upper bound of what reading scaled signals from the output buffer would buy: the forced-scaled headline tape with
the device-slot writes of values an OUTPUT_SCALED stores removed (results are wrong: readers find stale slots;
only the time means something)."""
import os, sys, struct, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import zkir_amd as z, bench
from zkir_amd import campaign

K = lambda w: int(w) >> 29
NONE = 5 << 29

def strip(t):
    ops = [list(op) for op in t.ops()]
    drop = set()
    removed = 0
    for i, o in enumerate(ops):
        if o[0] & 0xff != 9:
            continue
        j = i - 1
        while j >= 0:
            cj = ops[j][0] & 0xff
            if cj in (2, 9, 48):
                j -= 1; continue
            if K(ops[j][1]) == 0:
                removed += 1
                if cj == 3 and K(ops[j][2]) == 4:
                    drop.add(j)
                else:
                    ops[j][1] = NONE
            if cj == 3 and K(ops[j][2]) == 4:
                j -= 1; continue
            break
    kept = [op for k, op in enumerate(ops) if k not in drop]
    hdr = z.api.TapeHdr.from_buffer_copy(bytes(t.hdr))
    hdr.n_ops = len(kept)
    blob = b"".join(struct.pack("<4I", *op) for op in kept) + t.blob()[16 * t.hdr.n_ops:]
    print("removed", removed, "device-slot writes,", len(drop), "COPY ops; ops", t.hdr.n_ops, "->", len(kept))
    return z.Tape.from_parts(hdr, blob)

def timeit(ctx, tape, label, n=1 << 20, steps=5):
    dev = torch.device("cuda:0")
    h = tape.hdr
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    inp = campaign.uniform_inputs(torch, h.n_inputs, n, dev, gen)
    out = torch.empty((h.n_outputs, 8, n), dtype=torch.int32, device=dev)
    res = torch.zeros(4, dtype=torch.int64, device=dev)
    tid = ctx.upload(tape)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        ctx.eval_batch_dev(tid, inp.data_ptr(), n, out.data_ptr(), res.data_ptr(), s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.eval_batch_dev(tid, inp.data_ptr(), n, out.data_ptr(), res.data_ptr(), s)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print("%-40s %7.2f ms  ops %d gslots %d unsat %d" % (label, ms, h.n_ops, h.n_gslots, int(res[0])), flush=True)
    ctx.release(tid)

ctx = z.Context(0)
mode = sys.argv[1]
if mode == "plain":
    timeit(ctx, bench.workload_tape(z), "plain (cost rule)")
else:
    t = bench.workload_tape(z)
    timeit(ctx, t, "forced scaled")
    timeit(ctx, strip(t), "forced scaled, output-held values unwritten")
