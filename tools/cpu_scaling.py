#!/usr/bin/env python3
"""How the CPU checker (oracle/eval_oracle.c) scales over the host threads of this box: checks/s of the benchmark
circuit at 1, 2, 4, 8, 16 ... threads, for the tape as bench.py lowers it and for the same circuit in program order."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import zkir_amd as z  # noqa: E402
from oracle import c_oracle  # noqa: E402

rng = np.random.default_rng(1)
for name, env in (("as lowered for the GPU", {}), ("program order, no window cache", {"ZKIR_BENCH_REORDER": "0", "ZKIR_BENCH_WINDOW_CACHE": "0", "ZKIR_BENCH_ORDER_OPS": "0"})):
    os.environ.update(env)
    tape = bench.workload_tape(z)
    h = tape.hdr
    n = 16384
    inp = rng.integers(0, 1 << 32, size=(h.n_inputs, 8, n), dtype=np.uint64).astype(np.uint32)
    inp[:, 7, :] &= 0x0FFFFFFF
    for threads in (1, 2, 4, 8, 16, 32):
        if threads > (os.cpu_count() or 1):
            break
        t0 = time.perf_counter()
        _, unsat, _, _ = c_oracle.eval_cpu(bytes(h), tape.blob(), inp, n, h.n_outputs, want_outputs=True, threads=threads)
        dt = time.perf_counter() - t0
        print("%s: %d ops, %2d threads: %.4f G checks/s (%.2f s)" % (name, h.n_ops, threads, h.n_constraints * n / dt / 1e9, dt), flush=True)
print("cpus this process may use:", len(os.sched_getaffinity(0)), "of", os.cpu_count(), "OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"))
os.system("lscpu | grep -E 'Model name|Thread|Core|Socket|MHz' | head -8")
