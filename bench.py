#!/usr/bin/env python3
"""Headline benchmark: R1CS constraint checks per second over BN254 Fr.

One step = one pass of the hot path (zkir_eval_batch_dev: witness computation, output
write-back and constraint checks) over one batch of synthetic input vectors that is
already resident in device memory.  Workload: one generated circuit of about 1k
quadratic constraints evaluated for 2^20 input vectors per GPU (the second
configuration of BASELINE.json).  Vectors are independent, so with N GPUs every rank
evaluates its own 2^20 vectors (weak scaling: the headline `value`) and the only communication is one final
RCCL all-reduce of {unsatisfied pairs, checks done}.  The same line carries, under "also", the
fixed-size readings: configuration 2 and configuration 5 (the deep gadget, 2^16 vectors) with
the batch split over the GPUs (strong scaling), and a sample of configuration 4 (a campaign of
distinct circuits, circuit i on GPU i mod N, generate + lower + upload + evaluate + reduce).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus 4 ...            starts its own 4 ranks (one per GPU) when no launcher did
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


def workload_tape(z, signal_window=0, linear_terms=1, signal_factors=1, for_cpu=False):
    """configs[1] of BASELINE.json: about 1k rank-1 constraints `out <== lin*lin + lin`, about 1k
    products and about 2k additions/subtractions over 16 inputs.  linear_terms=1 (each `lin` is one
    signal or constant) gives that mix; 2 gives sums of two, i.e. twice the additions and more
    witness-slot traffic.  signal_factors=1: both factors of every constraint are signals, so every
    constraint is quadratic and its check multiplies; 0: a factor is a constant one time in four (the
    circuit of the earlier rounds' lines: 46 % of its constraints are `value * 1 == signal`)."""
    g = z.default_generator_config()
    g.signal_window = signal_window
    g.max_linear_terms = linear_terms
    g.signal_factors = signal_factors
    g.min_instructions = g.max_instructions = 4200
    g.forced_inputs, g.inputs_readable, g.arithmetic_only, g.quadratic_aware = 16, 1, 1, 1
    g.no_loops, g.safe_array_index = 1, 1
    o = z.default_lower_options()
    if for_cpu:
        # The same circuit as the CPU checker runs it fastest: statements in program order.  The order lower picks for
        # the GPU scatters the output stores over the output buffer, which costs the host cores their stride
        # prefetching (measured on the GPU box's 16 cores: 0.16 G checks/s in program order, 0.08 G in the GPU's order).
        o.reorder = o.window_cache = o.order_ops = o.scaled_values = 0
        return z.Lowerer(z.Generator(WORKLOAD_SEED, g).generate(), o).lower()
    for k in ("n_lslots", "window_lifetime", "use_acc", "reuse_slots", "strands", "max_xslots", "reorder", "window_cache", "order_ops", "scaled_values"):  # experiment knobs
        if os.environ.get("ZKIR_BENCH_" + k.upper()):
            setattr(o, k, int(os.environ["ZKIR_BENCH_" + k.upper()]))
    return z.Lowerer(z.Generator(WORKLOAD_SEED, g).generate(), o).lower()


def checks_by_one(tape):
    """How many CHECK ops of the tape have the constant 1 as right factor (`value * 1 == signal`, the constraint
    of a signal whose value is linear): the kernels compare those without a product, so they count as checks
    but not as field multiplications."""
    import numpy as np
    one = next((i for i, c in enumerate(tape.consts()) if c == 1), None)
    if one is None:
        return 0
    word = (2 << 29) | one  # operand kind CONST
    ops = np.frombuffer(tape.blob(), dtype=np.uint32, count=tape.hdr.n_ops * 4).reshape(-1, 4)
    return int(np.count_nonzero(((ops[:, 0] & 0xff) == 48) & (ops[:, 3] == word)))


def products_of(tape):
    """Field multiplications of the circuit per input vector: one per MUL op and one per constraint that multiplies.
    The products lower adds to move a value between its scaled and its plain form are work of the implementation,
    not of the circuit: they are not counted."""
    return tape.hdr.n_mul_ops - tape.hdr.n_scale_ops + tape.hdr.n_constraints - checks_by_one(tape)


def gadget_tape(z, rounds=20834, strands=0):
    """configs[4] of BASELINE.json: Poseidon-style chain, width 4, 3*4*rounds constraints.
    strands=4 lowers it into a strand tape (four wavefronts per 64 vectors): the shape for
    batches that leave the machine short of wavefronts."""
    o = z.default_lower_options()
    o.strands = strands
    for k in ("n_lslots", "window_lifetime", "use_acc", "reuse_slots", "strands", "max_xslots", "reorder", "window_cache", "order_ops", "scaled_values"):
        if os.environ.get("ZKIR_BENCH_" + k.upper()):
            setattr(o, k, int(os.environ["ZKIR_BENCH_" + k.upper()]))
    return z.Lowerer(z.poseidon_style_gadget(4, rounds, 7), o).lower()


def synthetic_inputs(torch, n_inputs, n, device, seed):
    """Field elements uniform in [0, p) (254 random bits, redrawn while not below p), limb-major SoA
    [n_inputs][8][n] of uint32."""
    from zkir_amd import campaign
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    return campaign.uniform_inputs(torch, n_inputs, n, device, gen)


def host_threads():
    """Host threads this process may use: the CPUs it is allowed on, capped by OMP_NUM_THREADS when the
    environment sets it (a GPU box shares its host: the variable holds one GPU's share of the cores)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    cap = os.environ.get("OMP_NUM_THREADS", "")
    if cap.isdigit() and int(cap) > 0:
        n = min(n, int(cap))
    return max(1, n)


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


def self_launch(args):
    """--gpus N without a launcher: start N ranks of this script as fresh child processes, one per GPU.
    The parent never touches the GPU (it does not even import torch); rank 0's child prints the line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(args.gpus))
    # RCCL shares device memory between the ranks' processes; the host driver of these machines only supports dmabuf
    # IPC, and with the legacy mode RCCL's set-up fails (hipIpcGetMemHandle: invalid argument).  A value the caller
    # has set is kept.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        for p in procs:
            rc = max(rc, abs(p.wait()))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2-vectors", type=int, default=LOG2_VECTORS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline measurement (no \"also\" entries)")
    ap.add_argument("--campaign-circuits", type=int, default=-1,
                    help="circuits of the configuration-4 sample under \"also\" (the whole job's, shared i mod N); by default "
                         "12 500 per GPU, which at 8 GPUs is the configuration's 100 000")
    ap.add_argument("--no-outputs", action="store_true",
                    help="config2 only: do not ask for the outputs (witness and checks only): a measurement aid")
    ap.add_argument("--workload", default="config2", choices=["config2", "gadget"],
                    help="config2 (default, the headline line) or gadget (the deep Poseidon-style chain, "
                         "2^16 vectors, outputs not requested)")
    ap.add_argument("--gadget-rounds", type=int, default=20834)
    ap.add_argument("--strong", action="store_true",
                    help="split the 2^log2-vectors batch over the ranks (contiguous ranges) instead of giving every "
                         "rank a batch of that size: the fixed-size reading of configs 2 and 5")
    ap.add_argument("--strands", default="auto", choices=["auto", "0", "4", "8"],
                    help="gadget only: lower into a strand tape of 4 or 8 strands (auto: 8 up to 2^14 vectors per GPU, else 4) or a plain tape (0)")
    ap.add_argument("--signal-window", type=int, default=0,
                    help="config2 only: linear terms read the inputs and the last W signals instead of any "
                         "earlier signal (0 = any, the default line)")
    ap.add_argument("--constant-factors", action="store_true",
                    help="config2 only: the circuit of the earlier rounds' lines, whose constraint factors are constants "
                         "one time in four (481 of its 1 047 constraints are `value * 1 == signal`)")
    ap.add_argument("--linear-terms", type=int, default=1, choices=[1, 2],
                    help="config2 only: terms per linear part of a constraint (1: the default line; 2: the "
                         "addition-heavy variant)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))

    import numpy as np
    import torch
    import torch.distributed as dist

    import zkir_amd as z

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # ZKIR_BENCH_REHEARSE=1: every rank on GPU 0 and gloo instead of RCCL, to walk the N > 1 control flow on a
    # one-GPU box (RCCL refuses two ranks on one device); the numbers of such a run mean nothing.
    # ZKIR_BENCH_DRYRUN=1: no GPU at all (gloo, no launches): the launcher, the barriers, the reductions and the
    # line's schema can be exercised on a CPU-only machine; the line says "dry_run" and its numbers mean nothing.
    dry = os.environ.get("ZKIR_BENCH_DRYRUN") == "1"
    rehearse = os.environ.get("ZKIR_BENCH_REHEARSE") == "1" or dry
    if rehearse:
        local_rank = 0
    if dry:
        device = torch.device("cpu")
    else:
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

    def fence():
        if world > 1:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    ctx = None if dry else z.Context(local_rank)
    stream = None if dry else torch.cuda.current_stream().cuda_stream

    def measure(tape, n, want_outputs, steps, warmup, seed):
        """W untimed and K timed launches of `tape` over this rank's n vectors, bracketed by barrier + device
        synchronisation; returns the whole job's totals and the MAX over ranks of the elapsed time."""
        h = tape.hdr
        tid = None if dry else ctx.upload(tape)
        inp = synthetic_inputs(torch, h.n_inputs, n, device, seed)
        out = torch.empty((h.n_outputs, 8, n), dtype=torch.int32, device=device) if want_outputs else None
        out_ptr = out.data_ptr() if out is not None else None
        res = torch.zeros(4, dtype=torch.int64, device=device)
        totals = torch.zeros(2, dtype=torch.int64, device=device)

        def launch():
            if not dry:
                ctx.eval_batch_dev(tid, inp.data_ptr(), n, out_ptr, res.data_ptr(), stream)

        totals[0] += res[0]  # the first use of these torch kernels loads them: not inside the timed region
        totals[1] += h.n_constraints * n
        for _ in range(warmup):
            launch()
        fence()
        totals.zero_()
        # per-launch kernel time on the launching stream (HIP events), beside the wall clock
        ev = [] if dry else [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        fence()
        t0 = time.perf_counter()
        for k in range(steps):
            if not dry:
                ev[k][0].record()
            launch()
            if not dry:
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
        kernel_ms = (elapsed * 1e3 / steps) if dry else sum(a.elapsed_time(b) for a, b in ev) / steps
        if not dry:
            ctx.release(tid)
        return {"unsat": unsat_total, "checks": checks_total, "elapsed": elapsed, "kernel_ms": kernel_ms, "inputs": inp}

    gadget = args.workload == "gadget"
    if gadget and args.log2_vectors == LOG2_VECTORS:
        args.log2_vectors = 16
    n = 1 << (10 if dry else args.log2_vectors)
    if args.strong:
        if n % world:
            raise SystemExit("--strong needs a batch the ranks divide evenly")
        n //= world  # this rank's contiguous range of the batch
    # measured (profiles/r01_f_gadget_strands_vs_plain.txt): the gadget's strand tape beats its plain tape at every
    # batch size from 2^13 to 2^21 vectors, by 4x where the plain tape leaves SIMDs without a wavefront
    # and (profiles/r02_d_strands8.txt) eight strands beat four up to 2^14 vectors per GPU (26.9 against 38.2 ms for 4 000 rounds)
    strands = (8 if n <= (1 << 14) else 4) if args.strands == "auto" else int(args.strands)
    tape = gadget_tape(z, args.gadget_rounds, strands) if gadget else \
        workload_tape(z, args.signal_window, args.linear_terms, 0 if args.constant_factors else 1)
    h = tape.hdr
    want_outputs = not gadget and not args.no_outputs
    m = measure(tape, n, want_outputs, args.steps, args.warmup, 1000 + rank)  # every rank its own vectors
    elapsed, kernel_ms, inp = m["elapsed"], m["kernel_ms"], m["inputs"]

    # ---- the fixed-size readings and the campaign sample ("also") -----------------------------------------
    also = []
    if not args.no_extras and not gadget and not args.strong and args.signal_window == 0 and args.linear_terms == 1 \
            and not args.no_outputs and args.log2_vectors == LOG2_VECTORS and not args.constant_factors:
        mulhi_peak_x = 1.0 if dry else ctx.int_mul_rate(0)

        def entry(name, tp, mm, n_rank, steps, scaling):
            hh = tp.hdr
            fr = products_of(tp) * n_rank
            return {"workload": name, "scaling": scaling, "value": mm["checks"] / mm["elapsed"], "unit": "checks/s",
                    "ms_per_step": 1e3 * mm["elapsed"] / steps, "kernel_ms": mm["kernel_ms"], "vectors_per_gpu": n_rank,
                    "n_constraints": hh.n_constraints, "strand_tape": bool(hh.flags & 1), "strands": hh.n_strands,
                    "fr_mul_per_s": fr * world / (mm["elapsed"] / steps),
                    "mulhi_frac": 128.0 * fr / (mm["kernel_ms"] * 1e-3) / mulhi_peak_x}

        # the circuit of the earlier rounds' headline lines, so that rounds stay comparable
        t_old = workload_tape(z, 0, 1, 0)
        mm = measure(t_old, n, True, args.steps, min(args.warmup, 2), 1500 + rank)
        assert mm["unsat"] == 0
        e_old = entry("configuration 2 with the earlier rounds' circuit (seed %d, a factor is a constant one time in four: "
                      "%d of its %d constraints are `value * 1 == signal` and are compared without a product) x 2^%d vectors "
                      "per GPU" % (WORKLOAD_SEED, checks_by_one(t_old), t_old.hdr.n_constraints, LOG2_VECTORS),
                      t_old, mm, n, args.steps, "weak")
        e_old["n_constraints_times_one"] = checks_by_one(t_old)
        also.append(e_old)
        if world > 1 and (1 << LOG2_VECTORS) % world == 0:
            n2 = (1 << (10 if dry else LOG2_VECTORS)) // world
            mm = measure(tape, n2, True, args.steps, min(args.warmup, 2), 2000 + rank)
            also.append(entry("configuration 2 as a fixed batch: the same circuit x 2^%d vectors split over the GPUs"
                              % LOG2_VECTORS, tape, mm, n2, args.steps, "strong"))
        if (1 << 16) % world == 0:
            g_rounds = 20 if dry else args.gadget_rounds
            n5 = (1 << (8 if dry else 16)) // world
            gt = gadget_tape(z, g_rounds, 8 if n5 <= (1 << 14) and not dry else 4)
            k5 = max(1, min(args.steps, 5))
            mm = measure(gt, n5, False, k5, 1, 3000 + rank)
            assert mm["unsat"] == 0
            also.append(entry("configuration 5: Poseidon-style gadget chain (width 4, %d rounds, %d constraints) x 2^16 "
                              "vectors split over the GPUs; witness + checks" % (g_rounds, gt.hdr.n_constraints),
                              gt, mm, n5, k5, "strong"))
        if args.campaign_circuits != 0:
            from zkir_amd import campaign
            n_circ, n_vec = (64 if dry else args.campaign_circuits if args.campaign_circuits > 0 else 12500 * world), (16 if dry else 1024)
            ids = campaign.shard(n_circ, rank, world)
            # the ranks of this job share one host (one node): every rank takes its share of the host's threads
            # (at least 4 where the machine has the CPUs: OMP_NUM_THREADS, which host_threads() honours, may hold one
            # GPU's share of the host rather than the job's)
            try:
                allowed = len(os.sched_getaffinity(0))
            except AttributeError:
                allowed = os.cpu_count() or 1
            budget = max(host_threads() // max(1, world), min(4, allowed // max(1, world)))
            workers = max(1, min(int(os.environ.get("ZKIR_CAMPAIGN_WORKERS", "16")), budget))
            if dry:
                def evaluator(tapes, n_vectors, circuit_ids):  # no GPU: the lowering and the exchange still run
                    return [0] * len(tapes), 0
            else:
                evaluator = campaign.GpuEvaluator(local_rank, input_seed=7 + rank)
            host_stats = {}
            fence()
            t0 = time.perf_counter()
            # a launch per batch the host hands over: the GPU has time to spare, and what is left to evaluate when the
            # host has finished is one small launch
            res_c = campaign.run_local(ids, 7, n_vec, evaluator, group=int(os.environ.get("ZKIR_CAMPAIGN_GROUP", "1024")),
                                       lowered=campaign.lower_parallel(ids, 7, workers=workers, stats=host_stats))
            t_red = time.perf_counter()
            tot = campaign.reduce_results(res_c, dist if world > 1 else None, device if not rehearse else None)
            fence()
            t_end = time.perf_counter()
            dt = torch.tensor([t_end - t0], dtype=torch.float64, device=device)
            if world > 1:
                all_reduce(dt, op=dist.ReduceOp.MAX)
            dt = float(dt.item())
            phases = {"generate_thread_s": host_stats.get("generate_s", 0.0), "lower_thread_s": host_stats.get("lower_s", 0.0),
                      "generate_lower_wall_s": host_stats.get("host_wall_s", 0.0), "reduce_s": t_end - t_red}
            if not dry:
                phases.update(evaluator.seconds)
                evaluator.close()
            also.append({"workload": "configuration 4 (sample): campaign of %d distinct circuits (64 to 4 096 constraints) x %d "
                                     "vectors each, circuit i on GPU i mod N; generate + lower + upload + evaluate + reduce"
                                     % (n_circ, n_vec),
                         "scaling": "strong", "value": tot["checks"] / dt, "unit": "checks/s", "seconds": dt,
                         "circuits_per_s": tot["circuits"] / dt, "circuits": tot["circuits"], "rejected": tot["rejected"],
                         "unsat": tot["unsat"], "faults": tot["faults"], "n_divergent": tot["n_divergent"],
                         "host_threads_per_rank": workers,
                         # rank 0's seconds per phase: generate and lower are thread-seconds over its workers (they run
                         # beside the evaluation on a helper thread, generate_lower_wall_s being that thread's wall clock);
                         # upload, inputs (sending the vector draws and the launch), wait (blocked on a group's results)
                         # and release are the evaluating thread's wall clock; evaluate is the GPU's own time over the
                         # groups (HIP events), which runs behind the upload of the next group
                         "phases_rank0": phases})

    if rank == 0:
        unsat_total, checks_total = m["unsat"], m["checks"]
        if world == 1:
            assert checks_total == h.n_constraints * n * args.steps
        assert unsat_total == 0, "synthetic witnesses must satisfy every constraint"
        value = checks_total / elapsed
        n_by_one = checks_by_one(tape)
        fr_muls = products_of(tape) * n  # per launch: one Montgomery product per Mul op and per constraint that multiplies
        mulhi_peak = 1.0 if dry else ctx.int_mul_rate(0)
        want = os.environ.get("ZKIR_INTERP", "")
        kernel = ("zkir_interp_kernel" if want == "compiled" else "zkir_interp_lean_kernel" if want == "lean32" else
                  "zkir_interp_lean29_kernel" if want == "lean29" else
                  "zkir_interp_lean_kernel" if h.n_gslots > 96 else "zkir_interp_lean29_kernel")  # the library's rule
        fr_peak = 1.0 if dry else ctx.fr_mul_rate(limbs29=kernel == "zkir_interp_lean29_kernel")
        achieved = 128.0 * fr_muls / (kernel_ms * 1e-3)
        alg_bytes = (32 * (h.n_inputs + (h.n_outputs if want_outputs else 0)) + 4) * n
        # HBM bytes per launch from the counter passes committed under profiles/ (only for the
        # exact workload they were taken on; counters cannot be read from inside this process)
        traffic = None
        if (want_outputs and args.signal_window == 0 and args.log2_vectors == LOG2_VECTORS and not args.strong
                and not gadget and not any(k.startswith("ZKIR_") for k in os.environ)):
            try:
                with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                    key = "config2%s_seed%d_2^%d_window4_terms%d" % ("" if args.constant_factors else "_signal_factors",
                                                                     WORKLOAD_SEED, LOG2_VECTORS, args.linear_terms)
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
                                (", constant factors one time in four" if args.constant_factors else ", both factors of every constraint signals") +
                                (", signal window %d" % args.signal_window if args.signal_window else ""),
                                args.log2_vectors, batch_of,
                                "witness + outputs + checks" if want_outputs else "witness + checks, outputs not requested")),
                "n_constraints": h.n_constraints, "n_constraints_times_one": n_by_one, "n_mul_ops": h.n_mul_ops - h.n_scale_ops,
                "n_scaling_products": h.n_scale_ops, "n_ops": h.n_ops,
                "n_inputs": h.n_inputs, "n_outputs": h.n_outputs, "vectors_per_gpu": n,
                "device_slots": h.n_gslots, "window_slots": h.n_lslots,
                "strand_tape": bool(h.flags & 1), "strands": h.n_strands, "exchange_slots": h.n_xslots,
            },
            "fr_mul_per_s": fr_muls * world / (1e3 * elapsed / args.steps * 1e-3),
            "roofline": {
                "kernel": kernel,
                "bound": "int32_mulhi",
                "achieved": achieved / 1e12,
                "peak": mulhi_peak / 1e12,
                "unit": "Tmulhi/s",
                "frac": achieved / mulhi_peak,
                "traffic": traffic,
                "peak_source": "measured on this device (v_mul_hi_u32 issue-rate probe); theoretical %.1f" % (MULHI_PEAK_THEORETICAL / 1e12),
                "frac_of_theoretical": achieved / MULHI_PEAK_THEORETICAL,
                "counted": "128 mul-hi per MUL op and per constraint with two signal factors; the %d constraints "
                           "`value * 1 == signal` are compared without a product and count as checks only" % n_by_one,
                # the counting of the first round's line (every constraint a product, also those against 1), for comparison only
                "frac_counting_every_constraint": 128.0 * (h.n_mul_ops - h.n_scale_ops + h.n_constraints) * n / (kernel_ms * 1e-3) / mulhi_peak,
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
        if also:
            line["also"] = also
        if dry:
            line["dry_run"] = True
        if world == 1 and not args.no_cpu_baseline and not dry:
            threads = host_threads()
            sample = min(n, (64 if gadget else 8192) * min(threads, 32))  # 10 to 30 s of CPU work in all
            inp_s = np.ascontiguousarray(inp[:, :, :sample].cpu().numpy().view(np.uint32))
            cpu_tape = tape if gadget else workload_tape(z, args.signal_window, args.linear_terms, 0 if args.constant_factors else 1, for_cpu=True)
            assert cpu_tape.hdr.n_constraints == h.n_constraints and cpu_tape.hdr.n_inputs == h.n_inputs
            cps, dt = cpu_baseline(cpu_tape, inp_s, threads)
            one = max(1, sample // 4)
            cps1, dt1 = cpu_baseline(cpu_tape, np.ascontiguousarray(inp_s[:, :, :one]), 1)
            line["cpu_baseline"] = {"value": cps, "unit": "checks/s", "cores": threads, "kind": "port",
                                    "host_cpus": os.cpu_count(),
                                    "sample": "first %d vectors of the same batch, the same circuit lowered in program order (the "
                                              "order the CPU runs fastest), %d threads (every core this process may use; the machine "
                                              "shows %s), %.1f s" % (sample, threads, os.cpu_count(), dt),
                                    "one_core": {"value": cps1, "sample": "first %d vectors, 1 thread, %.1f s" % (one, dt1)}}
        print(json.dumps(line), flush=True)
    if ctx is not None:
        ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
