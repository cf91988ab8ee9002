#!/bin/bash
# One GPU-box round trip: smoke, GPU parity tests, bench, rocprofv3 kernel stats.
# Usage (from the repo root on the box): bash scripts/gpu_check.sh [tag]
set -o pipefail
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -2 $OUT/smoke.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
timeout -k 10 600 python bench.py --steps 100 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -5 $OUT/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -12
