#!/bin/bash
# quick GPU round trip: build, GPU tests, short bench (no CPU baseline)
set -o pipefail
TAG=${1:-quick}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -8 $OUT/pytest_gpu.log
shift
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench.json")); print("value %.4g atom-steps/s  ms/step %.4f  kernel_ms %.4f  launch %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"]["launch"]))
except Exception as e:
    print("bench parse failed", e); print(open("$OUT/bench.err").read()[-2000:])
PY
