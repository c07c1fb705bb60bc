"""Fuzz-campaign driver: many generated circuits, a batch of random input vectors each,
sharded over the GPUs of a node.

The reference's own driver is `main()` (src/main.rs:81-92): seed, generate, hand the
circuit to a backend.  A campaign does the same per circuit id, with the seed derived
from (base seed, circuit id) so any circuit can be regenerated from its id alone.

Sharding: circuit i belongs to rank i mod world; ranks never exchange circuit data.
The only communication is at the end: one all-reduce of the counters and one
all-gather of the (padded) lists of divergent circuit ids.
"""
import math

import numpy as np

from . import api


def circuit_seed(base_seed, circuit_id):
    """Seed of circuit `circuit_id`: base + id + 1, never 0 (0 is the xorshift's fixed point)."""
    s = (base_seed + circuit_id + 1) & 0xFFFFFFFFFFFFFFFF
    return s or 0x9E3779B97F4A7C15


def shard(n_circuits, rank, world):
    """Circuit ids owned by `rank`."""
    return range(rank, n_circuits, world)


_DEFAULT_CONFIG_BYTES = None


def campaign_config(circuit_id, min_constraints=64, max_constraints=4096):
    """Generator knobs for campaign circuits: straight-line arithmetic circuits whose
    constraint count is spread log-uniformly over [min, max] by the circuit id."""
    global _DEFAULT_CONFIG_BYTES
    if _DEFAULT_CONFIG_BYTES is None:
        _DEFAULT_CONFIG_BYTES = bytes(api.default_generator_config())
    g = api.GeneratorConfig.from_buffer_copy(_DEFAULT_CONFIG_BYTES)  # (one library call per circuit adds up over a campaign)
    # about one statement in four is a constraint
    lo, hi = math.log(min_constraints * 4.0), math.log(max_constraints * 4.0)
    u = ((circuit_id * 2654435761) % 4096) / 4095.0
    n = int(round(math.exp(lo + (hi - lo) * u)))
    g.min_instructions = g.max_instructions = n
    g.forced_inputs, g.inputs_readable, g.arithmetic_only, g.quadratic_aware = 8, 1, 1, 1
    g.no_loops, g.safe_array_index = 1, 1
    return g


def campaign_lower_options():
    """How a campaign lowers: it is bound by the host side (a circuit costs 0.6 to 0.8 ms of one host thread to draw and
    lower, and 0.03 ms of the GPU to evaluate), so lower skips what only shortens the evaluation: the search for the
    statement order with the least device-memory traffic (2 ms per circuit), the choice of op order within a statement
    and the window run as a cache (0.08 ms each; measured on the GPU box's host: 0.755 -> 0.595 ms per circuit and thread
    without the two), and the choice of signals kept as scaled values (0.23 ms; the campaign launch runs at the rate HBM
    delivers, so the output conversions it saves buy no time there).
    ZKIR_CAMPAIGN_LOWER="order_ops=1,window_cache=1" in the environment sets options for a measurement."""
    import os
    o = api.default_lower_options()
    o.reorder = o.order_ops = o.window_cache = o.scaled_values = 0
    for item in filter(None, os.environ.get("ZKIR_CAMPAIGN_LOWER", "").split(",")):
        k, v = item.split("=")
        setattr(o, k.strip(), int(v))
    return o


def lower_parallel(circuit_ids, base_seed, workers=8, window=64, lower_options=None, config_of=campaign_config, stats=None):
    """Generate and lower campaign circuits on `workers` host threads; yields
    (circuit id, class, Tape or None) in id order.  The circuits go to the library `window` x workers
    at a time (zkir_generate_lower_many: the threads are the library's own and nothing passes through
    Python per circuit); one helper thread keeps the next batch in the making while the consumer works
    on the last one (ctypes drops the interpreter lock for the call, and nothing is forked from a process
    that may already hold a GPU context).  stats: optional dict; "host_wall_s" receives the seconds the helper
    thread spent inside the library, "generate_s" / "lower_s" the thread-seconds of the two phases over all workers.
    Closing the generator early (the consumer gave up) stops the helper thread at its next batch."""
    import queue
    import threading
    import time
    if lower_options is None:
        lower_options = campaign_lower_options()
    ids = list(circuit_ids)
    step = max(1, window * workers)
    batches = queue.Queue(maxsize=2)
    stop = threading.Event()
    g0, l0 = api.host_phase_seconds()

    def hand_over(item):
        """put that gives up when the consumer has gone (a bare put would block for ever and keep the tapes alive)"""
        while not stop.is_set():
            try:
                batches.put(item, timeout=0.2)
                return True
            except queue.Full:
                pass
        return False

    def produce():
        try:
            for at in range(0, len(ids), step):
                if stop.is_set():
                    return
                part = ids[at:at + step]
                t0 = time.perf_counter()
                made = api.generate_lower_many([circuit_seed(base_seed, c) for c in part], [config_of(c) for c in part],
                                               lower_options, workers)
                if stats is not None:
                    stats["host_wall_s"] = stats.get("host_wall_s", 0.0) + time.perf_counter() - t0
                if not hand_over((part, made)):
                    return
            hand_over(None)
        except BaseException as e:  # hand the failure to the consumer instead of leaving it waiting
            hand_over(e)

    threading.Thread(target=produce, daemon=True).start()
    try:
        while True:
            item = batches.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            part, made = item
            for cid, (cls, tape) in zip(part, made):
                yield cid, cls, tape
    finally:
        stop.set()
        if stats is not None:  # (process-wide counters: two campaigns lowering at once would see each other's time)
            g1, l1 = api.host_phase_seconds()
            stats["generate_s"], stats["lower_s"] = g1 - g0, l1 - l0


class LocalResult:
    """What one rank found: counters plus the ids it flags."""

    def __init__(self):
        self.circuits = 0       # circuits this rank owned
        self.rejected = 0       # refused by lower (counted per class in `classes`)
        self.checks = 0         # (vector, constraint) pairs evaluated
        self.unsat = 0          # pairs that did not hold
        self.faults = 0         # vectors that divided by zero
        self.classes = [0] * 9  # ZKIR_LOWER_* histogram
        self.divergent = []     # circuit ids with unsat pairs, faults, or an unexpected class


def run_local(circuit_ids, base_seed, n_vectors, evaluate, config_of=campaign_config, lower_options=None,
              expected_class=None, group=64, lowered=None):
    """Generate, lower and evaluate the circuits of one rank.

    lowered: optional iterator of (circuit id, class, Tape or None) for circuit_ids in order
    (lower_parallel makes one on several host cores); without it circuits are generated and
    lowered here, one after the other.

    evaluate(tapes, n_vectors, circuit_ids) -> (per-tape unsat list, faults) runs a group of
    lowered tapes for n_vectors fresh random vectors each; on a GPU rank it is
    GpuEvaluator below; an evaluator that also has submit(tapes, n_vectors, circuit_ids) -> handle with result() is
    run one group ahead (the next group is uploaded and launched before the last one's results are read).
    `faults` is the per-tape list of vectors that divided by zero (or, from an
    evaluator that only knows the launch's total, an integer: the group's first circuit is then flagged).  expected_class maps circuit id -> class name when an external
    verdict (e.g. circom's) is known; a different class makes the circuit divergent.
    """
    res = LocalResult()
    pending = []

    in_flight = []  # at most one group whose launch has been sent and whose results have not been read

    def account(ids, tapes, unsat, faults):
        per_tape = list(faults) if not isinstance(faults, int) else None
        res.faults += sum(per_tape) if per_tape is not None else faults
        for k, (cid, tape, u) in enumerate(zip(ids, tapes, unsat)):
            res.checks += tape.hdr.n_constraints * n_vectors
            res.unsat += u
            if u or (per_tape is not None and per_tape[k]):  # the circuit that failed a check or faulted
                res.divergent.append(cid)
        if per_tape is None and faults and not any(unsat):
            res.divergent.extend(ids[:1])  # only a total is known: flag the group's first circuit

    def collect():
        while in_flight:
            ids, tapes, handle = in_flight.pop(0)
            account(ids, tapes, *handle.result())

    def flush():
        if not pending:
            return
        ids = [i for i, _ in pending]
        tapes = [t for _, t in pending]
        pending.clear()
        submit = getattr(evaluate, "submit", None)
        if submit is None:
            account(ids, tapes, *evaluate(tapes, n_vectors, ids))
            return
        # an evaluator that can run behind the host: this group is uploaded and launched while the GPU still works
        # on the last one, whose results are read only then
        handle = submit(tapes, n_vectors, ids)
        collect()
        in_flight.append((ids, tapes, handle))

    def lower_here():
        for cid in circuit_ids:
            circuit = api.Generator(circuit_seed(base_seed, cid), config_of(cid)).generate()
            try:
                yield cid, 0, api.Lowerer(circuit, lower_options).lower()
            except api.LowerRejected as e:
                yield cid, e.status_class, None

    for cid, cls, tape in (lowered if lowered is not None else lower_here()):
        res.circuits += 1
        res.classes[cls] += 1
        if expected_class is not None and cid in expected_class and expected_class[cid] != api.LOWER_CLASSES[cls]:
            res.divergent.append(cid)
        if tape is None:
            res.rejected += 1
            continue
        pending.append((cid, tape))
        if len(pending) >= group:
            flush()
    flush()
    collect()
    res.divergent = sorted(set(res.divergent))
    return res


P_LIMBS = (0xf0000001, 0x43e1f593, 0x79b97091, 0x2833e848, 0x8181585d, 0xb85045b6, 0xe131a029, 0x30644e72)


def uniform_inputs(torch, n_elements, n_vectors, device, generator):
    """[n_elements][8][n_vectors] int32: field elements uniform in [0, p), limb-major.  Rejection sampling on the
    device: 254 random bits per candidate (p has 254 bits), kept when below p (three candidates in four are)."""
    need = n_elements * n_vectors
    if need == 0:
        return torch.zeros((max(n_elements, 1), 8, max(n_vectors, 1)), dtype=torch.int32, device=device)[:n_elements]
    kept, have = [], 0
    while have < need:
        m = int((need - have) * 1.36) + 4096
        x = torch.randint(0, 1 << 32, (8, m), dtype=torch.int64, device=device, generator=generator)
        x[7] &= 0x3FFFFFFF
        below = torch.zeros(m, dtype=torch.bool, device=device)
        equal = torch.ones(m, dtype=torch.bool, device=device)
        for limb in range(7, -1, -1):
            below |= equal & (x[limb] < P_LIMBS[limb])
            equal &= x[limb] == P_LIMBS[limb]
        good = x[:, below][:, :need - have].to(torch.int32)
        kept.append(good)
        have += good.shape[1]
    flat = kept[0] if len(kept) == 1 else torch.cat(kept, dim=1)
    return flat.view(8, n_elements, n_vectors).permute(1, 0, 2).contiguous()


class GpuEvaluator:
    """evaluate() for run_local on one MI355X: inputs are drawn on the device, every group of tapes is one launch
    (one tape per workgroup pass).  Launches go to a stream of their own, so that the upload of the next group (a
    synchronous copy on the null stream) runs while the GPU evaluates the last one: submit() sends a group off and
    returns a handle, result() of the handle waits for it, reads the counters back and releases the tapes."""

    class _Pending:
        def __init__(self, owner, tids, keep, unsat, faults, result, begin, end, order):
            self.owner, self.tids, self.keep, self.order = owner, tids, keep, order
            self.unsat, self.faults, self.result_words, self.begin, self.end = unsat, faults, result, begin, end

        def result(self):
            import time
            o = self.owner
            t0 = time.perf_counter()
            self.end.synchronize()
            t1 = time.perf_counter()
            sorted_out = [int(v) for v in self.unsat.cpu()], [int(v) for v in self.faults.cpu()]
            assert sum(sorted_out[1]) == int(self.result_words[2].item())
            out = [0] * len(self.order), [0] * len(self.order)  # back in the order the caller gave the tapes in
            for at, k in enumerate(self.order):
                out[0][k], out[1][k] = sorted_out[0][at], sorted_out[1][at]
            for tid in self.tids:
                o.ctx.release(tid)
            self.keep = None
            o.seconds["wait_s"] += t1 - t0
            o.seconds["evaluate_s"] += self.begin.elapsed_time(self.end) * 1e-3
            o.seconds["release_s"] += time.perf_counter() - t1
            return out

    def __init__(self, device_index, input_seed=1):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device_index)
        self.ctx = api.Context(device_index)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(input_seed)
        self.stream = torch.cuda.Stream(device=self.device)
        # host wall clock per phase, except evaluate_s: the GPU's time from the first input kernel of a group to the end
        # of its launch (HIP events); wait_s is what the host spent blocked on results
        self.seconds = {"upload_s": 0.0, "inputs_s": 0.0, "evaluate_s": 0.0, "wait_s": 0.0, "release_s": 0.0}

    def submit(self, tapes, n_vectors, circuit_ids):
        import time
        torch = self.torch
        t0 = time.perf_counter()
        # longest tapes first: workgroups draw (tape, pass) items in order, so the launch's tail is made of short ones
        order = sorted(range(len(tapes)), key=lambda k: -tapes[k].hdr.n_ops)
        tapes = [tapes[k] for k in order]
        tids = self.ctx.upload_many(tapes)
        t1 = time.perf_counter()
        with torch.cuda.stream(self.stream):
            begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            begin.record()
            # the tapes' blocks [n_inputs][8][n_vectors] lie one after the other: the whole buffer is [elements][8][n_vectors]
            inp = uniform_inputs(torch, max(1, sum(t.hdr.n_inputs for t in tapes)), n_vectors, self.device, self.gen)
            unsat = torch.zeros(len(tapes), dtype=torch.int64, device=self.device)
            faults = torch.zeros(len(tapes), dtype=torch.int64, device=self.device)
            result = torch.zeros(4, dtype=torch.int64, device=self.device)  # {0, first unsat pair, faults, reserved}
            self.ctx.eval_multi_dev(tids, inp.data_ptr(), n_vectors, unsat.data_ptr(), self.stream.cuda_stream, result.data_ptr(),
                                    faults.data_ptr())
            end.record()
        t2 = time.perf_counter()
        self.seconds["upload_s"] += t1 - t0
        self.seconds["inputs_s"] += t2 - t1
        return GpuEvaluator._Pending(self, tids, inp, unsat, faults, result, begin, end, order)

    def __call__(self, tapes, n_vectors, circuit_ids):
        return self.submit(tapes, n_vectors, circuit_ids).result()

    def close(self):
        self.ctx.close()


def reduce_results(res, dist=None, device=None, capacity=4096):
    """The campaign's single exchange step.  With torch.distributed initialised (RCCL on
    GPUs, gloo in the CPU tests) returns the job-wide totals and the sorted union of
    divergent ids on every rank; without it returns the local values.
    Ids beyond `capacity` per rank are dropped from the gather (the count still tells)."""
    import torch
    counters = [res.circuits, res.rejected, res.checks, res.unsat, res.faults, len(res.divergent)] + list(res.classes)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return {"circuits": res.circuits, "rejected": res.rejected, "checks": res.checks, "unsat": res.unsat,
                "faults": res.faults, "n_divergent": len(res.divergent), "classes": list(res.classes),
                "divergent": list(res.divergent)}
    world = dist.get_world_size()
    t = torch.tensor(counters, dtype=torch.int64, device=device)
    dist.all_reduce(t)  # sums: shards are disjoint
    ids = torch.full((capacity + 1,), -1, dtype=torch.int64, device=device)
    kept = res.divergent[:capacity]
    ids[0] = len(kept)  # the count travels first, the payload is padded
    if kept:
        ids[1:1 + len(kept)] = torch.tensor(kept, dtype=torch.int64, device=device)
    gathered = [torch.empty_like(ids) for _ in range(world)]
    dist.all_gather(gathered, ids)
    union = []
    for g in gathered:
        g = g.cpu()
        union.extend(int(v) for v in g[1:1 + int(g[0])])
    tot = [int(v) for v in t.cpu()]
    return {"circuits": tot[0], "rejected": tot[1], "checks": tot[2], "unsat": tot[3], "faults": tot[4],
            "n_divergent": tot[5], "classes": tot[6:], "divergent": sorted(set(union))}


def main(argv=None):
    """A differential campaign on the GPUs of one node: one rank per GPU under
    `python -m torch.distributed.run --nproc-per-node G -m zkir_amd.campaign ...`, or a single
    process.  Rank 0 prints one JSON line with the job-wide totals."""
    import argparse
    import json
    import os
    import time

    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--circuits", type=int, default=1024)
    ap.add_argument("--vectors", type=int, default=1024, help="input vectors per circuit")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--group", type=int, default=1024,
                    help="circuits per launch: a launch should hold several times the 1 280 workgroups the GPU runs at once "
                         "(1 024 circuits x 1 024 vectors are 4 096), while the next launch's tapes are lowered and uploaded "
                         "behind it (measured end to end on one GPU with 16 host threads: 512 -> 15 300 circuits/s, "
                         "1 024 -> 17 900, 2 048 -> 14 900)")
    ap.add_argument("--workers", type=int, default=0,
                    help="host threads that generate and lower, per rank (0: this process's CPUs divided by the ranks, at most 16)")
    ap.add_argument("--strands", type=int, default=0, choices=[0, 4, 8],
                    help="4: ask lower for strand tapes (circuits they do not fit come back as plain tapes); for "
                         "campaigns whose launches leave the GPU short of wavefronts")
    args = ap.parse_args(argv)

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)
    ev = GpuEvaluator(local_rank, input_seed=args.seed + rank)
    ids = shard(args.circuits, rank, world)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lower_options = None
    if args.strands:
        lower_options = api.default_lower_options()
        lower_options.strands = args.strands
    workers = args.workers
    if workers <= 0:
        try:
            allowed = len(os.sched_getaffinity(0))
        except AttributeError:
            allowed = os.cpu_count() or 1
        workers = max(1, min(16, allowed // max(1, world)))
    res = run_local(ids, args.seed, args.vectors, ev, group=args.group,
                    lowered=lower_parallel(ids, args.seed, workers=workers, lower_options=lower_options))
    tot = reduce_results(res, dist if world > 1 else None, device)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        tot.update({"n_gpus": world, "seconds": float(t.item()), "vectors_per_circuit": args.vectors,
                    "checks_per_s": tot["checks"] / float(t.item()), "circuits_per_s": tot["circuits"] / float(t.item()),
                    "classes": {api.LOWER_CLASSES[k]: v for k, v in enumerate(tot["classes"]) if v}})
        print(json.dumps(tot), flush=True)
    ev.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
