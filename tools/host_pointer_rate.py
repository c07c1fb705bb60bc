import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zkir_amd as z, bench
tape = bench.workload_tape(z)
h = tape.hdr
ctx = z.Context(0)
tid = ctx.upload(tape)
for logn in (14, 16, 17):
    n = 1 << logn
    rng = np.random.default_rng(1)
    inp = rng.integers(0, 1 << 32, size=(h.n_inputs, 8, n), dtype=np.uint64).astype(np.uint32)
    inp[:, 7, :] &= 0x0FFFFFFF
    for rep in range(2):
        t0 = time.perf_counter()
        out, unsat, first, faults = ctx.eval_batch(tid, inp, n)
        dt = time.perf_counter() - t0
    gb = (inp.nbytes + out.nbytes) / 1e9
    print("host-pointer entry: 2^%d vectors, %.3f s, %.2f G checks/s, %.1f GB moved -> %.1f GB/s" % (
        logn, dt, h.n_constraints * n / dt / 1e9, gb, gb / dt), flush=True)
    assert unsat == 0
    t0 = time.perf_counter()
    _, unsat, _, _ = ctx.eval_batch(tid, inp, n, want_outputs=False)
    dt = time.perf_counter() - t0
    print("   without outputs: %.3f s, %.2f G checks/s" % (dt, h.n_constraints * n / dt / 1e9), flush=True)
