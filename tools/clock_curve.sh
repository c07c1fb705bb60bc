#!/bin/bash
# Clock held under load (GRBM_GUI_ACTIVE over the kernel's duration, summed over the 8 XCDs) for variants of the
# headline launch that move different amounts of HBM traffic, with both generated streams.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/clkcurve
rm -rf $OUT; mkdir -p $OUT
run() {  # tag, env, flags...
  local tag=$1; local mode=$2; shift 2
  ZKIR_INTERP=$mode timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $OUT/$tag -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > $OUT/$tag.txt 2>&1
}
for mode in lean32 lean29; do
  run ${mode}_default $mode
  run ${mode}_nooutputs $mode --no-outputs
  run ${mode}_window8 $mode --signal-window 8
  run ${mode}_window8_nooutputs $mode --signal-window 8 --no-outputs
done
python3 - <<'PY'
import glob, os, sqlite3, json
out = "gpurun_out/clkcurve"
print("variant | kernel ms | GRBM_GUI_ACTIVE / 8 XCDs | clock GHz | checks/s")
for d in sorted(glob.glob(out + "/*/")):
    tag = os.path.basename(d.rstrip("/"))
    db = glob.glob(d + "*.db")[0]
    con = sqlite3.connect(db)
    cyc = con.execute("select avg(value) from counters_collection where kernel_name like '%interp%' and counter_name='GRBM_GUI_ACTIVE'").fetchone()[0]
    line = [l for l in open(out + "/" + tag + ".txt").read().splitlines() if l.startswith("{")][0]
    j = json.loads(line)
    ms = j["roofline"]["kernel_ms"]
    print("%-28s | %7.2f | %.4g | %.3f | %.3g" % (tag, ms, cyc / 8, cyc / 8 / (ms * 1e-3) / 1e9, j["value"]))
PY
rm -rf $OUT/*/
