"""The benchmark line committed under profiles/ (the output of `python bench.py` on an MI355X)
carries every field the driver's contract names, with consistent values."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


CURRENT_ROUND = 3


def latest_line():
    """The newest committed output of `python bench.py` on an MI355X: the last (by name) bench line of the current
    round under profiles/.  A round that commits no line of its own fails here: the checks below would otherwise
    keep passing on an old file."""
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r%02d_*bench*.json" % CURRENT_ROUND)))
    assert paths, "no bench line of round %d under profiles/" % CURRENT_ROUND
    with open(paths[-1]) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_bench_line_schema():
    d = latest_line()
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str),
                     ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # value is whole-job checks/s: constraints x vectors x steps over the timed region
    cfg = d["config"]
    per_step = cfg["n_constraints"] * cfg["vectors_per_gpu"] * d["n_gpus"]
    assert abs(d["value"] - per_step / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    # the counted products (bench.products_of): one per MUL op of the circuit and one per constraint that multiplies;
    # a constraint `value * 1 == signal` is compared without a product, a product lower added to rescale a value is
    # not the circuit's: 128 high products each
    fr = (cfg["n_mul_ops"] + cfg["n_constraints"] - cfg["n_constraints_times_one"]) * cfg["vectors_per_gpu"]
    assert abs(r["achieved"] * 1e12 - 128.0 * fr / (r["kernel_ms"] * 1e-3)) / (r["achieved"] * 1e12) < 1e-6
    # the headline circuit is the all-quadratic one: (next to) no constraint against the constant 1
    assert cfg["n_constraints_times_one"] <= 0.05 * cfg["n_constraints"]
    # the traffic figure is the counter passes' entry for this workload
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        t = json.load(f)
    assert r["traffic"] is not None
    assert any(abs(ent["bytes_per_launch"] - r["traffic"]) < 1 and ent["source"].startswith("profiles/r%02d_" % CURRENT_ROUND)
               for key, ent in t.items() if not key.startswith("_"))
    # the comparison entry keeps the earlier rounds' circuit, the campaign sample says where its seconds went
    also = {e["workload"].split(":")[0].split(" with ")[0].split(" as ")[0] + ("/old" if " with the earlier" in e["workload"] else ""): e
            for e in d["also"]}
    assert also["configuration 2/old"]["n_constraints_times_one"] > 100
    assert set(also["configuration 4 (sample)"]["phases_rank0"]) >= {"generate_thread_s", "lower_thread_s", "upload_s", "evaluate_s", "reduce_s"}


def test_traffic_file_matches_profiles():
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        t = json.load(f)
    for key, ent in t.items():
        if key.startswith("_"):
            continue
        assert os.path.exists(os.path.join(ROOT, ent["source"])), ent["source"]
        want = ent["fetch_size_kb_raw"] * 2048 + ent["write_size_kb"] * 1024
        assert abs(ent["bytes_per_launch"] - want) / want < 1e-6


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher starts two ranks itself and rank 0 prints one line.  Here
    without a GPU (ZKIR_BENCH_DRYRUN=1: gloo, no launches), so only the launcher, the barriers, the reductions,
    the campaign sample's sharding and the line's shape are exercised."""
    import subprocess
    import sys
    env = dict(os.environ, ZKIR_BENCH_DRYRUN="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    cfg = d["config"]
    assert abs(d["value"] - cfg["n_constraints"] * cfg["vectors_per_gpu"] * 2 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    kinds = [e["workload"].split(":")[0].split(" as ")[0].split(" with ")[0] for e in d["also"]]
    assert kinds == ["configuration 2", "configuration 2", "configuration 5", "configuration 4 (sample)"]
    assert [e["scaling"] for e in d["also"]] == ["weak", "strong", "strong", "strong"] and all(e["value"] > 0 for e in d["also"])
    assert d["config"]["n_constraints_times_one"] == 0 and d["also"][0]["n_constraints_times_one"] > 100
    camp = d["also"][3]
    assert camp["circuits"] == 64 and camp["unsat"] == 0  # both ranks' shares were counted once each
    assert camp["host_threads_per_rank"] >= 1 and set(camp["phases_rank0"]) >= {"generate_thread_s", "lower_thread_s", "reduce_s"}


def test_products_are_mul_ops_plus_constraints_that_multiply():
    """bench.py prices a launch at 128 mul-hi per field multiplication: one per MUL op and one per constraint with two
    signal factors.  A constraint `value * 1 == signal` is compared without a product (the kernels do that, and the
    count follows); the gadget has none."""
    import bench
    import zkir_amd as z
    assert bench.checks_by_one(bench.workload_tape(z)) == 0  # the headline circuit: every constraint has two signal factors
    t = bench.workload_tape(z, signal_factors=0)               # the earlier rounds' circuit
    ops, consts = t.ops(), t.consts()
    by_one = sum(1 for code, _, _, b in ops if code & 0xFF == 48 and b >> 29 == 2 and consts[b & 0x1FFFFFFF] == 1)
    assert by_one > 100 and bench.checks_by_one(t) == by_one
    # (the products lower adds to rescale values are not the circuit's)
    assert t.hdr.n_scale_ops > 100 and bench.products_of(t) == t.hdr.n_mul_ops - t.hdr.n_scale_ops + t.hdr.n_constraints - by_one
    assert bench.workload_tape(z).hdr.n_scale_ops == 0  # the all-quadratic circuit stays plain: copies would outnumber its savings
    g = bench.gadget_tape(z, 5, 4)
    assert bench.checks_by_one(g) == 0 and bench.products_of(g) == g.hdr.n_mul_ops + g.hdr.n_constraints
