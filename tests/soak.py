"""Differential soak: random circuits under random generator / lowering options, GPU vs the C checker."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zkir_amd as z
from zkir_amd import api
from oracle import c_oracle as C
from oracle import fr_ref as F

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
BIG = os.environ.get("SOAK_BIG") == "1"  # circuits of the benchmark's size and beyond: tapes that spill heavily
ctx = z.Context(0)
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 20240611)  # python tests/soak.py [iterations [seed]]
special = [0, 1, 2, F.P - 1, F.P - 2, F.HALF, F.HALF + 1, (1 << 64) - 1, 1 << 253, 253, 254, 255, F.P - 253, F.P - 254]
done = rejected = bad = full = multi = stranded = scaled = 0
group = []
t0 = time.time()
for it in range(N):
    g = z.default_generator_config()
    n = rnd.choice([320, 640, 1280, 2560, 5120] if BIG else [20, 40, 80, 160, 320, 640])
    g.min_instructions = g.max_instructions = n
    g.max_expr_depth = rnd.choice([1, 1, 2, 3])
    g.inputs_readable = 1
    g.forced_inputs = rnd.choice([0, 1, 3, 8])
    g.arithmetic_only = rnd.choice([0, 1, 1])
    g.quadratic_aware = rnd.choice([0, 1, 1])
    g.no_loops = rnd.choice([0, 1, 1])
    g.safe_array_index = 1
    g.enable_if = rnd.choice([0, 0, 2])
    g.max_linear_terms = rnd.choice([0, 1])
    g.signal_window = rnd.choice([0, 0, 4])
    g.signal_factors = rnd.choice([0, 1])
    o = z.default_lower_options()
    o.n_lslots = rnd.choice([0, 1, 2, 3, 4, 4, 8, 16])
    o.use_acc = rnd.choice([0, 1, 1, 1])
    o.reuse_slots = rnd.choice([0, 1, 1])
    o.window_lifetime = rnd.choice([4, 16, 48, 200])
    o.fold_constants = rnd.choice([0, 1])
    o.bnot_variant = rnd.choice([0, 1])
    o.max_iterations = 300
    o.strands = rnd.choice([0, 0, 4, 8])  # strand tapes where the circuit allows (+ - * only, few crossing values)
    o.max_xslots = rnd.choice([0, 0, 6, 16])
    o.reorder = rnd.choice([0, 1, 1])
    o.window_cache = rnd.choice([0, 1, 1])
    o.order_ops = rnd.choice([0, 1, 1])
    o.scaled_values = rnd.choice([0, 1, 1])  # (the pass keeps its rewrite only where it pays: ZKIR_SCALE_CONVERT_COST=0.1 in the environment makes it keep almost all)
    seed = rnd.randrange(1, 1 << 62)
    try:
        tape = z.Lowerer(z.Generator(seed, g).generate(), o).lower()
    except api.LowerRejected:
        rejected += 1
        continue
    nv = rnd.choice([1, 63, 64, 65, 200, 257, 513])
    vecs = [[rnd.choice(special) if rnd.random() < 0.3 else rnd.randrange(F.P) for _ in range(tape.hdr.n_inputs)]
            for _ in range(nv)]
    inp = z.pack_inputs(vecs, tape.hdr.n_inputs)
    for mode in ("lean29", "lean32", "compiled"):
        if mode:
            os.environ["ZKIR_INTERP"] = mode
        else:
            os.environ.pop("ZKIR_INTERP", None)
        tid = ctx.upload(tape)
        out, unsat, first, faults = ctx.eval_batch(tid, inp, nv)
        ctx.release(tid)
        o_c, u_c, first_c, f_c = C.eval_cpu(bytes(tape.hdr), tape.blob(), inp, nv, tape.hdr.n_outputs)
        if not np.array_equal(out, o_c) or (unsat, first, faults) != (u_c, first_c, f_c):
            bad += 1
            print("MISMATCH seed", seed, "mode", mode or "default", "cfg", [getattr(g, k) for k, _ in g._fields_][:18],
                  "opts", [getattr(o, k) for k, _ in o._fields_], "nv", nv, flush=True)
    # every so often the recent tapes also go through one multi-tape launch, one of them broken
    group.append(tape)
    if len(group) == 12:
        nv2 = rnd.choice([1, 64, 130, 300])
        k = rnd.randrange(len(group))
        ops = group[k].ops()
        muls = [i for i, op in enumerate(ops) if (op[0] & 0xFF) == F.OP_MUL]
        if muls:  # turn a product into a sum: constraints that use it stop holding (most of the time)
            j = rnd.choice(muls)
            ops[j] = (F.OP_ADD,) + ops[j][1:]
            import struct
            blob = b"".join(struct.pack("<4I", *op) for op in ops) + group[k].blob()[16 * len(ops):]
            group[k] = z.Tape.from_parts(group[k].hdr, blob)
        blocks, want = [], []
        for t in group:
            v2 = [[rnd.choice(special) if rnd.random() < 0.3 else rnd.randrange(F.P) for _ in range(t.hdr.n_inputs)]
                  for _ in range(nv2)]
            i2 = z.pack_inputs(v2, t.hdr.n_inputs)
            blocks.append(i2.reshape(-1))
            want.append(C.eval_cpu(bytes(t.hdr), t.blob(), i2, nv2, t.hdr.n_outputs, want_outputs=False)[1])
        # strand tapes and plain tapes share the call (two launches behind it)
        tids = ctx.upload_many(group)
        got, _ = ctx.eval_multi(tids, np.concatenate(blocks) if blocks else None, nv2)
        for t in tids:
            ctx.release(t)
        if got != want:
            bad += 1
            print("MISMATCH in a multi-tape launch at iteration", it, got, want, flush=True)
        multi += 1
        group = []
    done += 1
    stranded += tape.hdr.flags & 1
    codes = {c & 0xFF for c, _, _, _ in tape.ops()}
    full += bool(codes - {0, 1, 2, 3, 4, 7, 8, 9, 16, 17, 18, 48})
    scaled += bool(codes & {8, 9})
    if it % 100 == 0:
        print("iter", it, "evaluated", done, "rejected", rejected, "mismatches", bad, "%.0fs" % (time.time() - t0), flush=True)
os.environ.pop("ZKIR_INTERP", None)
print("SOAK done: evaluated %d circuits (%d with non-arithmetic operators, %d as strand tapes, %d with scaled values) and %d "
      "multi-tape launches, rejected %d, mismatches %d" % (done, full, stranded, scaled, multi, rejected, bad))
sys.exit(1 if bad else 0)
