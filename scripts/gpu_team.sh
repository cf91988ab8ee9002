#!/bin/bash
# compare the wavefront-per-atom and workgroup-per-atom kernels (tests run with the default = team when available)
OUT=gpurun_out/${1:-team}
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
for cfg in "0 0" "1 1" "1 2" "1 3" "1 4"; do
  set -- $cfg
  MTP_TEAM=$1 MTP_TEAM_PER_CU=$2 timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/b_$1_$2.json 2> $OUT/b_$1_$2.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/b_$1_$2.json")); print("TEAM=$1 PER_CU=$2 ms/step %.4f kernel_ms %.4f launch %s" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"]["launch"]))
except Exception as e:
    print("TEAM=$1 PER_CU=$2 failed", e)
PY
done
for cells in 8 16; do
for t in 0 1; do
  MTP_TEAM=$t timeout -k 10 200 python bench.py --cells $cells --steps 200 --warmup 20 --no-cpu-baseline > $OUT/s_${cells}_$t.json 2> $OUT/s_${cells}_$t.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/s_${cells}_$t.json")); print("cells=$cells TEAM=$t atoms %d ms/step %.4f value %.4g" % (d["config"]["atoms"], d["ms_per_step"], d["value"]))
except Exception as e:
    print("cells=$cells TEAM=$t failed", e)
PY
done; done
