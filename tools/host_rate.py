#!/usr/bin/env python3
"""Host side of a campaign alone (no GPU work): thread-milliseconds per circuit of zkir_generate_lower_many on campaign
circuits, for the placement options a campaign may lower with.  usage: host_rate.py [circuits] [threads]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zkir_amd import api, campaign  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ids = list(range(n))
seeds = [campaign.circuit_seed(7, c) for c in ids]
cfgs = [campaign.campaign_config(c) for c in ids]


def run(**kw):
    o = api.default_lower_options()
    o.reorder = 0
    for k, v in kw.items():
        setattr(o, k, v)
    g0, l0 = api.host_phase_seconds()
    t0 = time.perf_counter()
    made = api.generate_lower_many(seeds, cfgs, o, threads)
    dt = time.perf_counter() - t0
    g1, l1 = api.host_phase_seconds()
    print("%-40s %d threads: %.3f ms wall per circuit = %.0f circuits/s; thread-ms per circuit: draw + pass 1 %.3f, placement %.3f; "
          "device slots %.0f, ops %.0f per tape" % (kw, threads, 1e3 * dt / n, n / dt, 1e3 * (g1 - g0) / n, 1e3 * (l1 - l0) / n,
                                                   sum(t.hdr.n_gslots for _, t in made) / n, sum(t.hdr.n_ops for _, t in made) / n), flush=True)


run()
for rep in range(2):
    run()
    run(order_ops=0)
    run(window_cache=0)
    run(window_cache=0, order_ops=0)
