"""Soak of the fused generate + lower (no GPU): zkir_generate_lower_many on straight-line configurations must give,
byte for byte, the header, blob and io maps (or the refusal class) of Generator(seed, cfg).generate() followed by
Lowerer(circuit, opts).lower().  Usage: python tests/soak_fused.py [circuits [seed]]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkir_amd as z

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 31)
t0 = time.time()
done = refused = bad = 0
while done < N:
    seeds, cfgs = [], []
    for k in range(200):
        g = z.default_generator_config()
        g.min_instructions = rnd.choice([0, 1, 5, 40, 150, 600, 2000])
        g.max_instructions = g.min_instructions + rnd.choice([0, 0, 7, 300])
        g.no_loops, g.quadratic_aware, g.arithmetic_only = 1, 1, 1
        g.forced_inputs = rnd.choice([0, 0, 1, 8, 16])
        g.inputs_readable = rnd.choice([0, 1, 1])
        g.safe_array_index = rnd.choice([0, 1, 1])
        g.signal_window = rnd.choice([0, 0, 3, 8])
        g.max_linear_terms = rnd.choice([0, 1, 2, 3])
        g.signal_factors = rnd.choice([0, 1])
        g.max_array_len = rnd.choice([1, 2, 10, 64])
        g.statement_budget = rnd.choice([10, 10, 1, 3, 50])
        g.name_len = rnd.choice([8, 1, 30])
        seeds.append(rnd.getrandbits(64) | 1)
        cfgs.append(g)
    o = z.default_lower_options()
    o.max_iterations = 300
    o.reorder, o.window_cache, o.order_ops, o.scaled_values = (rnd.choice([0, 1]) for _ in range(4))
    o.n_lslots = rnd.choice([0, 2, 4])
    o.strands = rnd.choice([0, 0, 0, 4, 8])
    o.fold_constants = rnd.choice([0, 1])
    got = z.generate_lower_many(seeds, cfgs, o, 8)
    for k, (cls, t) in enumerate(got):
        try:
            w = z.Lowerer(z.Generator(seeds[k], cfgs[k]).generate(), o).lower()
            want = (0, bytes(w.hdr), w.blob(), w.io_map(0), w.io_map(1))
        except z.LowerRejected as e:
            want = (e.status_class, None, None, None, None)
            refused += 1
        have = (cls, bytes(t.hdr) if t else None, t.blob() if t else None, t.io_map(0) if t else None, t.io_map(1) if t else None)
        if have != want:
            bad += 1
            print("MISMATCH seed", seeds[k], "cfg", [getattr(cfgs[k], f) for f, _ in cfgs[k]._fields_][:19], "opts", [getattr(o, f) for f, _ in o._fields_], flush=True)
        done += 1
    print("%d circuits (%d refused), %d mismatches, %.0f s" % (done, refused, bad, time.time() - t0), flush=True)
print("soak of the fused generate + lower: %d circuits (%d refused by lower), %d mismatches" % (done, refused, bad))
sys.exit(1 if bad else 0)
