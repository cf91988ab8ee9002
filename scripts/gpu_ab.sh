#!/bin/bash
# A/B of tuning knobs on the headline workload: bash scripts/gpu_ab.sh TAG "ENV1=a ENV2=b" "ENV1=c" ...
# (each quoted argument is one configuration's environment; "-" = defaults)
set -o pipefail
TAG=${1:-ab}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
i=0
for cfg in "$@"; do
  i=$((i+1))
  envs=""; [ "$cfg" != "-" ] && envs="$cfg"
  env $envs timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $BENCH_ARGS > $OUT/b$i.json 2> $OUT/b$i.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/b$i.json")); print("[$cfg] ms/step %.4f kernel_ms %.4f launch %s" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"]["launch"]))
except Exception as e:
    print("[$cfg] failed", e); print(open("$OUT/b$i.err").read()[-1500:])
PY
done
