#!/bin/bash
# rocprofv3 kernel stats of one bench workload: bash scripts/gpu_stats_wl.sh TAG [bench args]
OUT=gpurun_out/${1:-stats}; shift
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-whole-step "$@" > $OUT/b.json 2> $OUT/b.err
find $OUT/st -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/stats.csv
rm -rf $OUT/st
grep "mtp_" $OUT/stats.csv | sed 's/void (anonymous namespace):://;s/(anonymous namespace):://;s/(MtpDevParams)//;s/([^"]*)"/"/' | cut -d, -f1-4
python - <<PY
import json
d=json.load(open("$OUT/b.json")); print("ms/step %.4f kernel_ms %.4f" % (d["ms_per_step"], d["roofline"]["kernel_ms"]))
PY
