"""The multi-rank path of the campaign driver on CPU: two gloo ranks shard the circuits
(i mod 2), evaluate them with the C checker standing in for the GPU, and exchange only
the final counters and divergent ids; the result must equal a single-rank run."""
import json
import os
import subprocess
import sys

import numpy as np

import zkir_amd as z
from oracle import c_oracle as C
from zkir_amd import campaign

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_CIRCUITS, N_VECTORS, BASE_SEED = 24, 16, 99


def small_config(cid):
    g = campaign.campaign_config(cid, min_constraints=4, max_constraints=40)
    if cid % 5 == 0:  # some circuits keep the reference's loops so lower refuses a few
        g.no_loops, g.quadratic_aware = 0, 0
    return g


def oracle_evaluate(tapes, n_vectors, ids):
    """Stand-in for GpuEvaluator in CPU tests: the C checker, plus a planted divergence."""
    unsat = []
    for cid, t in zip(ids, tapes):
        rng = np.random.default_rng(cid)
        inp = rng.integers(0, 1 << 32, size=(t.hdr.n_inputs, 8, n_vectors), dtype=np.uint64).astype(np.uint32)
        inp[:, 7, :] &= 0x0FFFFFFF
        _, u, _, _ = C.eval_cpu(bytes(t.hdr), t.blob(), inp, n_vectors, t.hdr.n_outputs, want_outputs=False)
        unsat.append(u + (3 if cid % 7 == 3 else 0))
    return unsat, 0


def run_rank(rank, world):
    o = z.default_lower_options()
    o.max_iterations = 200
    res = campaign.run_local(campaign.shard(N_CIRCUITS, rank, world), BASE_SEED, N_VECTORS, oracle_evaluate,
                             config_of=small_config, lower_options=o, group=5)
    return res


def test_an_evaluator_that_runs_one_group_ahead_gives_the_same_campaign():
    """run_local hands an evaluator with submit() the next group before it reads the last group's results (what
    GpuEvaluator does with launches): same totals, same divergent ids, and the calls come in that order."""
    log = []

    class Ahead:
        def __call__(self, tapes, n_vectors, ids):
            raise AssertionError("an evaluator with submit() is driven through it")

        def submit(self, tapes, n_vectors, ids):
            log.append(("submit", ids[0]))
            outer = self

            class Handle:
                def result(self):
                    log.append(("result", ids[0]))
                    return oracle_evaluate(tapes, n_vectors, ids)
            return Handle()

    o = z.default_lower_options()
    o.max_iterations = 200
    want = campaign.reduce_results(run_rank(0, 1))
    got = campaign.reduce_results(campaign.run_local(campaign.shard(N_CIRCUITS, 0, 1), BASE_SEED, N_VECTORS, Ahead(),
                                                     config_of=small_config, lower_options=o, group=5))
    assert got == want
    kinds = [k for k, _ in log]
    assert kinds[:3] == ["submit", "submit", "result"] and kinds[-1] == "result"
    assert kinds.count("submit") == kinds.count("result") >= 3
    # every group's results are read exactly once, in the order the groups were sent
    assert [i for k, i in log if k == "submit"] == [i for k, i in log if k == "result"]


def test_shard_is_a_partition():
    for world in (1, 2, 3, 8):
        got = sorted(i for r in range(world) for i in campaign.shard(37, r, world))
        assert got == list(range(37))
    assert campaign.circuit_seed(2**64 - 1, 0) != 0 and campaign.circuit_seed(5, 1) == 7


def test_two_gloo_ranks_equal_one_rank(tmp_path):
    single = campaign.reduce_results(run_rank(0, 1))
    assert single["circuits"] == N_CIRCUITS and single["checks"] > 0
    script = tmp_path / "rank.py"
    script.write_text(
        "import sys, json, os\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import torch.distributed as dist\n"
        "import test_campaign_cpu as T\n"
        "from zkir_amd import campaign\n"
        "dist.init_process_group('gloo')\n"
        "res = T.run_rank(dist.get_rank(), dist.get_world_size())\n"
        "tot = campaign.reduce_results(res, dist=dist, capacity=8)\n"
        "tot['local_circuits'] = res.circuits\n"
        "open(os.path.join(%r, 'out%%d.json' %% dist.get_rank()), 'w').write(json.dumps(tot))\n"
        "dist.destroy_process_group()\n" % (ROOT, os.path.join(ROOT, "tests"), str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    outs = [json.loads((tmp_path / ("out%d.json" % r)).read_text()) for r in range(2)]
    assert outs[0]["local_circuits"] + outs[1]["local_circuits"] == N_CIRCUITS
    for o in outs:
        o.pop("local_circuits")
        assert o == single  # every rank holds the job-wide totals and the union of divergent ids
    assert single["unsat"] == 3 * len(single["divergent"]) and len(single["divergent"]) >= 2
    assert single["rejected"] == sum(single["classes"][1:]) and sum(single["classes"]) == N_CIRCUITS


def test_batched_generate_and_lower_equals_one_by_one():
    """zkir_generate_lower_many on several threads gives, circuit for circuit, the tape (or the refusal)
    that zkir_generate + zkir_lower give, and lower_parallel hands them out in id order."""
    o = z.default_lower_options()
    o.max_iterations = 200
    ids = list(range(40))
    expect = []
    for cid in ids:
        c = z.Generator(campaign.circuit_seed(BASE_SEED, cid), small_config(cid)).generate()
        try:
            t = z.Lowerer(c, o).lower()
            expect.append((0, bytes(t.hdr), t.blob(), t.io_map(0), t.io_map(1)))
        except z.LowerRejected as e:
            expect.append((e.status_class, None, None, None, None))
    assert any(cls for cls, *_ in expect) and any(cls == 0 for cls, *_ in expect)
    for threads in (1, 3, 64):
        got = z.generate_lower_many([campaign.circuit_seed(BASE_SEED, c) for c in ids], [small_config(c) for c in ids], o, threads)
        assert [(cls, bytes(t.hdr) if t else None, t.blob() if t else None, t.io_map(0) if t else None,
                 t.io_map(1) if t else None) for cls, t in got] == expect
    out = list(campaign.lower_parallel(ids, BASE_SEED, workers=2, window=3, lower_options=o, config_of=small_config))
    assert [cid for cid, _, _ in out] == ids
    assert [(cls, t.blob() if t else None) for _, cls, t in out] == [(e[0], e[2]) for e in expect]
    assert z.generate_lower_many([], None, None, 4) == []
    # a bad generator config fails the whole call with the library's error, not a crash or a partial list
    bad = small_config(1)
    bad.min_instructions, bad.max_instructions = 9, 3
    try:
        z.generate_lower_many([5, 6, 7], [small_config(0), bad, small_config(2)], o, 2)
        assert False, "expected an error"
    except z.ZkirError as e:
        assert e.code == z.E_INVALID
    try:
        z.generate_lower_many([5, 0], None, o, 2)  # seed 0 is rejected like zkir_generate rejects it
        assert False, "expected an error"
    except z.ZkirError as e:
        assert e.code == z.E_INVALID


def test_circuits_lowered_while_generated_give_the_tape_of_generate_then_lower():
    """Straight-line campaign configurations never build the Circuit: zkir_generate_lower_many hands every statement
    to lower as it is drawn.  Same draws, same tape bytes, same io maps, same refusals as Generator(seed, cfg).generate()
    followed by Lowerer(circuit, opts).lower(), over seeds, sizes, every knob such a configuration has and the
    placement options; configurations that need the circuit (loops, If, deep expressions, every operator) still
    go through it and agree as well."""
    import random
    rnd = random.Random(5)
    seeds, cfgs = [], []
    for k in range(260):
        g = z.default_generator_config()
        g.min_instructions = rnd.choice([0, 1, 5, 40, 150, 600])
        g.max_instructions = g.min_instructions + rnd.choice([0, 0, 7, 300])
        g.no_loops, g.quadratic_aware, g.arithmetic_only = 1, 1, 1
        g.forced_inputs = rnd.choice([0, 0, 1, 8])
        g.inputs_readable = rnd.choice([0, 1, 1])
        g.safe_array_index = rnd.choice([0, 1, 1])
        g.signal_window = rnd.choice([0, 0, 3])
        g.max_linear_terms = rnd.choice([0, 1, 2])
        g.signal_factors = rnd.choice([0, 1])
        g.max_array_len = rnd.choice([1, 10, 64])
        g.statement_budget = rnd.choice([10, 10, 1, 3])
        g.name_len = rnd.choice([8, 1, 30])
        if k % 13 == 0:    # these need the circuit built first
            g.no_loops = 0
        elif k % 13 == 1:
            g.arithmetic_only = 0
        elif k % 13 == 2:
            g.max_expr_depth = 2
        elif k % 13 == 3:
            g.enable_if = 2
        seeds.append(rnd.choice([1, 2, 3, 2**64 - 1, rnd.getrandbits(64) | 1]))
        cfgs.append(g)
    n_refused = n_tapes = 0
    for variant in range(3):
        o = z.default_lower_options()
        o.max_iterations = 300
        if variant == 1:
            o.reorder, o.window_cache, o.order_ops, o.n_lslots = 0, 0, 0, 4
        if variant == 2:
            o.strands, o.reorder = 4, 0
        expect = []
        for s, g in zip(seeds, cfgs):
            c = z.Generator(s, g).generate()
            try:
                t = z.Lowerer(c, o).lower()
                expect.append((0, bytes(t.hdr), t.blob(), t.io_map(0), t.io_map(1)))
                n_tapes += 1
            except z.LowerRejected as e:
                expect.append((e.status_class, None, None, None, None))
                n_refused += 1
        got = z.generate_lower_many(seeds, cfgs, o, 4)
        for k, ((cls, t), want) in enumerate(zip(got, expect)):
            have = (cls, bytes(t.hdr) if t else None, t.blob() if t else None, t.io_map(0) if t else None, t.io_map(1) if t else None)
            assert have == want, (variant, k, seeds[k], cls, want[0])
    assert n_refused > 20 and n_tapes > 300
    # the switch that turns the short cut off (a test aid) gives the same tapes as well
    import subprocess
    import sys
    code = ("import zkir_amd as z\nfrom zkir_amd import campaign\nimport hashlib\n"
            "ids=list(range(60))\nmade=z.generate_lower_many([campaign.circuit_seed(3,c) for c in ids],[campaign.campaign_config(c,16,200) for c in ids],None,2)\n"
            "print(hashlib.sha256(b''.join(t.blob() for _,t in made if t)).hexdigest())")
    import os
    outs = []
    for env in ({}, {"ZKIR_NO_FUSED": "1"}):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert p.returncode == 0, p.stderr[-1000:]
        outs.append(p.stdout.strip())
    assert outs[0] == outs[1] and len(outs[0]) == 64
