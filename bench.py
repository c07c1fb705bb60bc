#!/usr/bin/env python3
"""Headline benchmark: R1CS constraint checks per second over BN254 Fr.

One step = one pass of the hot path (zkir_eval_batch_dev: witness computation, output
write-back and constraint checks) over one batch of synthetic input vectors that is
already resident in device memory.  Workload: one generated circuit of about 1k
quadratic constraints evaluated for 2^20 input vectors per GPU (the second
configuration of BASELINE.json).  Vectors are independent, so with N GPUs every rank
evaluates its own 2^20 vectors (weak scaling) and the only communication is one final
RCCL all-reduce of {unsatisfied pairs, checks done}.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD_SEED = 2024
LOG2_VECTORS = 20
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, datasheet
MULHI_PEAK_THEORETICAL = 256 * 4 * 16 * 2.4e9  # CUs x SIMDs x lanes/clk x max clock: one v_mul_hi_u32 per lane per clock


def workload_tape(z, signal_window=0, linear_terms=1):
    """configs[1] of BASELINE.json: about 1k rank-1 constraints `out <== lin*lin + lin`, about 1k
    products and about 2k additions/subtractions over 16 inputs.  linear_terms=1 (each `lin` is one
    signal or constant) gives that mix; 2 gives sums of two, i.e. twice the additions and more
    witness-slot traffic."""
    g = z.default_generator_config()
    g.signal_window = signal_window
    g.max_linear_terms = linear_terms
    g.min_instructions = g.max_instructions = 4200
    g.forced_inputs, g.inputs_readable, g.arithmetic_only, g.quadratic_aware = 16, 1, 1, 1
    g.no_loops, g.safe_array_index = 1, 1
    o = z.default_lower_options()
    for k in ("n_lslots", "window_lifetime", "use_acc", "reuse_slots", "strands", "max_xslots"):  # experiment knobs
        if os.environ.get("ZKIR_BENCH_" + k.upper()):
            setattr(o, k, int(os.environ["ZKIR_BENCH_" + k.upper()]))
    return z.Lowerer(z.Generator(WORKLOAD_SEED, g).generate(), o).lower()


def gadget_tape(z, rounds=20834, strands=0):
    """configs[4] of BASELINE.json: Poseidon-style chain, width 4, 3*4*rounds constraints.
    strands=4 lowers it into a strand tape (four wavefronts per 64 vectors): the shape for
    batches that leave the machine short of wavefronts."""
    o = z.default_lower_options()
    o.strands = strands
    for k in ("n_lslots", "window_lifetime", "use_acc", "reuse_slots", "strands", "max_xslots"):
        if os.environ.get("ZKIR_BENCH_" + k.upper()):
            setattr(o, k, int(os.environ["ZKIR_BENCH_" + k.upper()]))
    return z.Lowerer(z.poseidon_style_gadget(4, rounds, 7), o).lower()


def synthetic_inputs(torch, n_inputs, n, device, seed):
    """Uniform 252-bit values (every one below p), limb-major SoA [n_inputs][8][n] of uint32."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    x = torch.randint(0, 1 << 32, (n_inputs, 8, n), dtype=torch.int64, device=device, generator=gen).to(torch.int32)
    x[:, 7, :] &= 0x0FFFFFFF
    return x.contiguous()


def cpu_baseline(tape, inputs_sample, threads):
    """The scalar C interpreter of oracle/ on the host cores, on a bounded sample."""
    from oracle import c_oracle
    n = inputs_sample.shape[2]
    t0 = time.perf_counter()
    _, unsat, _, _ = c_oracle.eval_cpu(bytes(tape.hdr), tape.blob(), inputs_sample, n, tape.hdr.n_outputs,
                                       want_outputs=True, threads=threads)
    dt = time.perf_counter() - t0
    assert unsat == 0
    return tape.hdr.n_constraints * n / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2-vectors", type=int, default=LOG2_VECTORS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-outputs", action="store_true",
                    help="config2 only: do not ask for the outputs (witness and checks only): a measurement aid")
    ap.add_argument("--workload", default="config2", choices=["config2", "gadget"],
                    help="config2 (default, the headline line) or gadget (the deep Poseidon-style chain, "
                         "2^16 vectors, outputs not requested)")
    ap.add_argument("--gadget-rounds", type=int, default=20834)
    ap.add_argument("--strong", action="store_true",
                    help="split the 2^log2-vectors batch over the ranks (contiguous ranges) instead of giving every "
                         "rank a batch of that size: the fixed-size reading of configs 2 and 5")
    ap.add_argument("--strands", default="auto", choices=["auto", "0", "4"],
                    help="gadget only: lower into a strand tape (4, and auto) or a plain tape (0)")
    ap.add_argument("--signal-window", type=int, default=0,
                    help="config2 only: linear terms read the inputs and the last W signals instead of any "
                         "earlier signal (0 = any, the default line)")
    ap.add_argument("--linear-terms", type=int, default=1, choices=[1, 2],
                    help="config2 only: terms per linear part of a constraint (1: the default line; 2: the "
                         "addition-heavy variant)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import zkir_amd as z

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one rank per GPU with torch.distributed.run" % (args.gpus, world))
    # ZKIR_BENCH_REHEARSE=1: every rank on GPU 0 and gloo instead of RCCL, to walk the N > 1 control flow on a
    # one-GPU box (RCCL refuses two ranks on one device); the numbers of such a run mean nothing
    rehearse = os.environ.get("ZKIR_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    def all_reduce(t, op=dist.ReduceOp.SUM):
        if rehearse:  # gloo: reduce on the host
            c = t.cpu()
            dist.all_reduce(c, op=op)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=op)

    gadget = args.workload == "gadget"
    if gadget and args.log2_vectors == LOG2_VECTORS:
        args.log2_vectors = 16
    n = 1 << args.log2_vectors
    if args.strong:
        if n % world:
            raise SystemExit("--strong needs a batch the ranks divide evenly")
        n //= world  # this rank's contiguous range of the batch
    # measured (profiles/r01_f_gadget_strands_vs_plain.txt): the gadget's strand tape beats its plain tape at every
    # batch size from 2^13 to 2^21 vectors, by 4x where the plain tape leaves SIMDs without a wavefront
    strands = 4 if args.strands == "auto" else int(args.strands)
    tape = gadget_tape(z, args.gadget_rounds, strands) if gadget else workload_tape(z, args.signal_window, args.linear_terms)
    h = tape.hdr
    ctx = z.Context(local_rank)
    tid = ctx.upload(tape)
    inp = synthetic_inputs(torch, h.n_inputs, n, device, 1000 + rank)  # every rank its own vectors
    want_outputs = not gadget and not args.no_outputs
    out = torch.empty((h.n_outputs, 8, n), dtype=torch.int32, device=device) if want_outputs else None
    out_ptr = out.data_ptr() if out is not None else None
    res = torch.zeros(4, dtype=torch.int64, device=device)
    totals = torch.zeros(2, dtype=torch.int64, device=device)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        ctx.eval_batch_dev(tid, inp.data_ptr(), n, out_ptr, res.data_ptr(), stream)
        totals[0] += res[0]
        totals[1] += h.n_constraints * n

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    totals.zero_()
    # per-launch kernel time on the launching stream (HIP events), beside the wall clock
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        ctx.eval_batch_dev(tid, inp.data_ptr(), n, out_ptr, res.data_ptr(), stream)
        ev[k][1].record()
        totals[0] += res[0]
        totals[1] += h.n_constraints * n
    if world > 1:
        all_reduce(totals)  # the path's only exchange: mismatch count and checks done
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    unsat_total, checks_total = (int(v) for v in totals.cpu())
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps

    if rank == 0:
        if world == 1:
            assert checks_total == h.n_constraints * n * args.steps
        assert unsat_total == 0, "synthetic witnesses must satisfy every constraint"
        value = checks_total / elapsed
        fr_muls = (h.n_mul_ops + h.n_constraints) * n  # per launch: one Montgomery product per Mul op and per check
        mulhi_peak = ctx.int_mul_rate(0)
        fr_peak = ctx.fr_mul_rate()
        achieved = 128.0 * fr_muls / (kernel_ms * 1e-3)
        alg_bytes = (32 * (h.n_inputs + (h.n_outputs if want_outputs else 0)) + 4) * n
        # HBM bytes per launch from the counter passes committed under profiles/ (only for the
        # exact workload they were taken on; counters cannot be read from inside this process)
        traffic = None
        if (want_outputs and args.signal_window == 0 and args.log2_vectors == LOG2_VECTORS and not args.strong
                and not any(k.startswith("ZKIR_") for k in os.environ)):
            try:
                with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                    key = "config2_seed%d_2^%d_window4_terms%d" % (WORKLOAD_SEED, LOG2_VECTORS, args.linear_terms)
                    traffic = json.load(f)[key]["bytes_per_launch"]
            except (OSError, KeyError, ValueError):
                traffic = None
        batch_of = "split over the GPUs" if args.strong else "per GPU"
        line = {
            "metric": "R1CS constraint checks/sec over BN254 Fr",
            "value": value,
            "unit": "checks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": ("Poseidon-style gadget chain (width 4, %d rounds) x 2^%d vectors %s; witness + checks, "
                             "outputs not requested" % (args.gadget_rounds, args.log2_vectors, batch_of)) if gadget else
                            ("one generated circuit (extended generator, seed %d%s) x 2^%d uniform input vectors %s; %s" % (
                                WORKLOAD_SEED,
                                (", %d-term linear parts" % args.linear_terms) +
                                (", signal window %d" % args.signal_window if args.signal_window else ""),
                                args.log2_vectors, batch_of,
                                "witness + outputs + checks" if want_outputs else "witness + checks, outputs not requested")),
                "n_constraints": h.n_constraints, "n_mul_ops": h.n_mul_ops, "n_ops": h.n_ops,
                "n_inputs": h.n_inputs, "n_outputs": h.n_outputs, "vectors_per_gpu": n,
                "device_slots": h.n_gslots, "window_slots": h.n_lslots,
                "strand_tape": bool(h.flags & 1), "exchange_slots": h.n_xslots,
            },
            "fr_mul_per_s": fr_muls * world / (1e3 * elapsed / args.steps * 1e-3),
            "roofline": {
                "kernel": ("zkir_interp_kernel" if os.environ.get("ZKIR_INTERP") == "compiled"
                           else "zkir_interp_lean_kernel"),
                "bound": "int32_mulhi",
                "achieved": achieved / 1e12,
                "peak": mulhi_peak / 1e12,
                "unit": "Tmulhi/s",
                "frac": achieved / mulhi_peak,
                "traffic": traffic,
                "peak_source": "measured on this device (v_mul_hi_u32 issue-rate probe); theoretical %.1f" % (MULHI_PEAK_THEORETICAL / 1e12),
                "frac_of_theoretical": achieved / MULHI_PEAK_THEORETICAL,
                "kernel_ms": kernel_ms,
                "fr_mul_per_s_kernel": fr_muls / (kernel_ms * 1e-3),
                "fr_mul_per_s_register_ceiling": fr_peak,
            },
            "roofline_hbm": {
                "bound": "hbm", "achieved": alg_bytes / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_GBps": (traffic / (kernel_ms * 1e-3) / 1e9) if traffic else None,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            sample = min(n, (64 if gadget else 8192) * threads)  # about 20 core-seconds of work in all
            inp_s = np.ascontiguousarray(inp[:, :, :sample].cpu().numpy().view(np.uint32))
            cps, dt = cpu_baseline(tape, inp_s, threads)
            one = max(1, sample // 4)
            cps1, dt1 = cpu_baseline(tape, np.ascontiguousarray(inp_s[:, :, :one]), 1)
            line["cpu_baseline"] = {"value": cps, "unit": "checks/s", "cores": threads, "kind": "port",
                                    "sample": "first %d vectors of the same batch, %d threads, %.1f s" % (sample, threads, dt),
                                    "one_core": {"value": cps1, "sample": "first %d vectors, 1 thread, %.1f s" % (one, dt1)}}
        print(json.dumps(line), flush=True)
    ctx.release(tid)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
