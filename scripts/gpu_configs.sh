#!/bin/bash
# bench lines for the BASELINE.json configurations other than the headline one + N>1 rehearsal
OUT=gpurun_out/${1:-configs}
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_multirank_gpu.py -x -q -m gpu > $OUT/pytest_multirank.log 2>&1; echo "multirank pytest rc=$?"; tail -4 $OUT/pytest_multirank.log
for wl in small2k wre20 grades; do
  timeout -k 10 400 python bench.py --workload $wl --steps 50 --warmup 5 --cpu-seconds 6 > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; echo "$wl rc=$?"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$wl.json")); c=d.get("cpu_baseline") or {}
    print("$wl: value %.4g atom-steps/s  ms/step %.4f  kernel_ms %.4f  cpu %.4g  launch %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], c.get("value",0), d["config"]["launch"]))
except Exception as e:
    print("$wl parse failed", e); print(open("$OUT/bench_$wl.err").read()[-1500:])
PY
done
MTP_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29733 bench.py --gpus 2 --steps 20 --warmup 3 > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err; echo "2-rank rehearsal rc=$?"; tail -c 600 $OUT/bench_2rank_gloo.json
