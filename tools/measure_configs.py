#!/usr/bin/env python3
"""Measures the BASELINE.json configurations that are not the headline line:
  config 3: many distinct generated circuits x 1024 vectors each, one launch per group
  config 5: the deep Poseidon-style gadget x 2^16 vectors
Run on a GPU box:  python tools/measure_configs.py [n_circuits]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import zkir_amd as z
    from zkir_amd import campaign

    n_circuits = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    n_vec = 1024
    dev = torch.device("cuda:0")
    ctx = z.Context(0)
    t0 = time.time()
    # the tapes run many times here, so lower may search for the statement order with the least device-memory traffic
    # (ZKIR_MEASURE_REORDER=0: program order, as a campaign lowers them)
    options = z.default_lower_options()
    options.reorder = int(os.environ.get("ZKIR_MEASURE_REORDER", "1"))
    options.scaled_values = int(os.environ.get("ZKIR_MEASURE_SCALED", str(options.scaled_values)))
    options.n_lslots = int(os.environ.get("ZKIR_MEASURE_LSLOTS", str(options.n_lslots)))  # window rows (0: lower's choice)
    tapes = [t for _, cls, t in campaign.lower_parallel(range(n_circuits), 7, workers=16, lower_options=options) if t is not None]
    # longest tapes first: the work queue hands items out in order, so the tail is made of short ones
    tapes.sort(key=lambda t: -t.hdr.n_ops)
    host_s = time.time() - t0
    tids = [ctx.upload(t) for t in tapes]
    words = sum(t.hdr.n_inputs for t in tapes) * 8 * n_vec
    gen = torch.Generator(device=dev)
    gen.manual_seed(11)
    inp = campaign.uniform_inputs(torch, max(1, words // (8 * n_vec)), n_vec, dev, gen).view(-1)  # uniform in [0, p)
    unsat = torch.zeros(len(tapes), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    group = int(os.environ.get('ZKIR_GROUP', len(tids)))
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.time()
        for g in range(0, len(tids), group):
            ids = tids[g:g + group]
            in_off = sum(t.hdr.n_inputs for t in tapes[:g]) * 8 * n_vec * 4
            ctx.eval_multi_dev(ids, inp.data_ptr() + in_off, n_vec, unsat.data_ptr() + 8 * g, stream)
        torch.cuda.synchronize()
        dt = time.time() - t0
    checks = sum(t.hdr.n_constraints for t in tapes) * n_vec
    import bench  # products per vector: MUL ops + constraints that multiply (a check against the constant 1 compares only)
    muls = sum(bench.products_of(t) for t in tapes) * n_vec
    print(json.dumps({"config": "3: %d distinct circuits (%d lowered OK, %d..%d constraints) x %d vectors" % (
        n_circuits, len(tapes), min(t.hdr.n_constraints for t in tapes), max(t.hdr.n_constraints for t in tapes), n_vec),
        "gpu_ms": 1e3 * dt, "checks_per_s": checks / dt, "fr_mul_per_s": muls / dt, "unsat": int(unsat.sum()),
        "host_generate_lower_s_16_threads": host_s, "reorder": options.reorder, "scaled_values": options.scaled_values, "window_rows": max(t.hdr.n_lslots for t in tapes),
        "device_slots_max": max(t.hdr.n_gslots for t in tapes),
        "scaling_products": sum(t.hdr.n_scale_ops for t in tapes),
        "mulhi_frac": 128.0 * muls / dt / ctx.int_mul_rate(0)}))


if __name__ == "__main__":
    main()
