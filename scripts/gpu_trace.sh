#!/bin/bash
# kernel timeline of a few bench steps: bash scripts/gpu_trace.sh TAG [ENV=..] -- [bench args]
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
envs=""; while [ "$1" != "--" ] && [ -n "$1" ]; do envs="$envs $1"; shift; done; shift
env $envs timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-whole-step "$@" > $OUT/bench.json 2> $OUT/bench.err
echo rc=$?
python - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last 2 steps: print the last 40 kernels with start offsets
t0 = int(rows[-60]["Start_Timestamp"]) if len(rows) > 60 else int(rows[0]["Start_Timestamp"])
for r in rows[-60:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%7.1f us  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
PY
