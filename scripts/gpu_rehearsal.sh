#!/bin/bash
# The decomposed step on ONE GPU (self halo; default one-stream schedule and the overlapped one) next to the plain step
# at 8192 / 16000 / 31250 / 65536 atoms:
#   bash scripts/gpu_rehearsal.sh r02   -> gpurun_out/<tag>_profiles/<tag>_self_halo_rehearsal.json (best of 2 runs each)
TAG=${1:-r02}
OUT=gpurun_out/${TAG}_profiles
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
echo '[' > $OUT/${TAG}_self_halo_rehearsal.json
first=1
for c in 16 20 25 32; do
  for mode in plain self self_overlap; do
    envs="A=1"; [ $mode = self ] && envs="MTP_BENCH_SELF_HALO=1"; [ $mode = self_overlap ] && envs="MTP_BENCH_SELF_HALO=1 MTP_BENCH_HALO_OVERLAP=1"
    for rep in 1 2; do
      env $envs timeout -k 10 200 python bench.py --cells $c --steps 300 --warmup 20 --no-cpu-baseline --no-whole-step > $OUT/sh$rep.json 2> $OUT/sh.err
    done
    [ $first = 1 ] || echo ',' >> $OUT/${TAG}_self_halo_rehearsal.json
    first=0
    python - >> $OUT/${TAG}_self_halo_rehearsal.json <<PY
import json
ds = [json.load(open("$OUT/sh%d.json" % r)) for r in (1, 2)]
d = min(ds, key=lambda x: x["ms_per_step"])
print(json.dumps({"cells": $c, "atoms": d["config"]["atoms"], "mode": "$mode", "ms_per_step": d["ms_per_step"], "ms_per_step_runs": [x["ms_per_step"] for x in ds], "kernel_ms_single_launch": d["roofline"]["kernel_ms"], "parallelism": d["config"]["parallelism"]}))
PY
  done
done
echo ']' >> $OUT/${TAG}_self_halo_rehearsal.json
rm -f $OUT/sh1.json $OUT/sh2.json
cat $OUT/${TAG}_self_halo_rehearsal.json
