"""Soak of the lowering alone (no GPU): random circuits under random generator and lowering options; the tape with
its statements reordered, its window run as a cache, its values scaled where lower chooses, or cut into strands must give, vector for vector, the outputs,
the unsatisfied count and the first unsatisfied pair of the same circuit's tape in program order (C checker), also
with one constraint broken.  Usage: python tests/soak_lower.py [circuits [seed]]
(ZKIR_SCALE_CONVERT_COST=0.1 in the environment makes lower keep scaled values almost wherever it can place them.)"""
import os, random, struct, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zkir_amd as z
from zkir_amd import api
from oracle import c_oracle as C
from oracle import fr_ref as F

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 77)
special = [0, 1, 2, F.P - 1, F.P - 2, F.HALF, F.HALF + 1, (1 << 64) - 1, 1 << 253]
done = rejected = bad = reordered = copies = scaled = 0
t0 = time.time()


def broken(t, number):
    """(header, blob) of the tape with constraint `number` comparing its product with zero.  The zero is a constant
    appended to the pool: zero is zero whether lower kept the compared value plain or scaled, which no constant
    already in the pool need be."""
    ops, k = [list(op) for op in t.ops()], 0
    zero = F.operand(F.K_CONST, t.hdr.n_consts)
    for op in ops:
        if op[0] & 0xFF == F.OP_CHECK:
            if ((op[0] >> 8) - 1 if op[0] >> 8 else k) == number:
                op[1] = zero
            k += 1
    blob = t.blob()
    consts_end = 16 * t.hdr.n_ops + 32 * t.hdr.n_consts
    hdr = api.TapeHdr.from_buffer_copy(bytes(t.hdr))
    hdr.n_consts += 1
    return bytes(hdr), b"".join(struct.pack("<4I", *op) for op in ops) + blob[16 * t.hdr.n_ops:consts_end] + bytes(32) + blob[consts_end:]


for it in range(N):
    g = z.default_generator_config()
    g.min_instructions = g.max_instructions = rnd.choice([20, 40, 80, 160, 320, 640, 1500])
    g.max_expr_depth = rnd.choice([1, 1, 2, 3])
    g.inputs_readable = 1
    g.forced_inputs = rnd.choice([0, 1, 3, 8, 16])
    g.arithmetic_only = rnd.choice([0, 1, 1])
    g.quadratic_aware = rnd.choice([0, 1, 1])
    g.no_loops = rnd.choice([0, 1, 1])
    g.safe_array_index = 1
    g.enable_if = rnd.choice([0, 0, 2])
    g.max_linear_terms = rnd.choice([0, 1])
    g.signal_window = rnd.choice([0, 0, 4])
    g.signal_factors = rnd.choice([0, 1])
    seed = rnd.randrange(1, 1 << 62)
    c = z.Generator(seed, g).generate()
    base = z.default_lower_options()
    base.reorder = base.window_cache = base.order_ops = base.scaled_values = 0
    base.max_iterations = 300
    base.fold_constants = rnd.choice([0, 1])
    base.bnot_variant = rnd.choice([0, 1])
    try:
        ref = z.Lowerer(c, base).lower()
    except api.LowerRejected:
        rejected += 1
        continue
    nv = rnd.choice([1, 3, 9])
    vecs = [[rnd.choice(special) if rnd.random() < 0.3 else rnd.randrange(F.P) for _ in range(ref.hdr.n_inputs)] for _ in range(nv)]
    inp = z.pack_inputs(vecs, ref.hdr.n_inputs)
    want = C.eval_cpu(bytes(ref.hdr), ref.blob(), inp, nv, ref.hdr.n_outputs)
    number = rnd.randrange(ref.hdr.n_constraints) if ref.hdr.n_constraints else None
    want_bad = C.eval_cpu(*broken(ref, number), inp, nv, ref.hdr.n_outputs) if number is not None else None
    for trial in range(3):
        o = z.default_lower_options()
        o.max_iterations, o.fold_constants, o.bnot_variant = 300, base.fold_constants, base.bnot_variant
        o.n_lslots = rnd.choice([0, 1, 2, 3, 4, 8, 16])
        o.use_acc = rnd.choice([0, 1, 1, 1])
        o.reuse_slots = rnd.choice([0, 1, 1])
        o.window_lifetime = rnd.choice([4, 48, 200])
        o.strands = rnd.choice([0, 0, 0, 4, 8])
        o.reorder = rnd.choice([0, 1, 1])
        o.window_cache = rnd.choice([0, 1, 1])
        o.order_ops = rnd.choice([0, 1, 1])
        o.scaled_values = rnd.choice([0, 1, 1])
        t = z.Lowerer(c, o).lower()
        scaled += t.hdr.n_scale_ops > 0 or any(op[0] & 0xFF in (8, 9) for op in t.ops())
        z.Tape.from_parts(t.hdr, t.blob())
        reordered += any(op[0] >> 8 for op in t.ops()) and not t.hdr.flags
        copies += sum(1 for op in t.ops() if op[0] & 0xFF == F.OP_COPY and op[2] >> 29 == F.K_ACC)
        got = C.eval_cpu(bytes(t.hdr), t.blob(), inp, nv, t.hdr.n_outputs)
        ok = np.array_equal(got[0], want[0]) and got[1:] == want[1:]
        if ok and number is not None:
            gb = C.eval_cpu(*broken(t, number), inp, nv, t.hdr.n_outputs)
            ok = np.array_equal(gb[0], want_bad[0]) and gb[1:] == want_bad[1:]
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, "cfg", [getattr(g, k) for k, _ in g._fields_][:18], "opts", [getattr(o, k) for k, _ in o._fields_], flush=True)
    done += 1
    if done % 200 == 0:
        print("%d circuits (%d refused), %d reordered tapes, %d device copies, %d tapes with scaled values, %d mismatches, %.0f s" % (done, rejected, reordered, copies, scaled, bad, time.time() - t0), flush=True)
print("soak of the lowering: %d circuits x 3 option sets (%d refused by lower), %d reordered plain tapes, %d device copies of window values, %d tapes with scaled values, %d mismatches, %.0f s"
      % (done, rejected, reordered, copies, scaled, bad, time.time() - t0))
sys.exit(1 if bad else 0)
