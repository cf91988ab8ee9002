#!/bin/bash
# tuning sweep of the LDS tile / workgroup shape (benchmarks only)
OUT=gpurun_out/${1:-sweep}
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
for cfg in "32 1" "32 2" "32 4" "32 8" "32 7" "32 6"; do
  set -- $cfg
  MTP_NT=$1 MTP_WPB=$2 timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/b_$1_$2.json 2> $OUT/b_$1_$2.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/b_$1_$2.json")); print("NT=$1 WPB=$2 ms/step %.4f kernel_ms %.4f launch %s" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"]["launch"]))
except Exception as e:
    print("NT=$1 WPB=$2 failed", e)
PY
done
