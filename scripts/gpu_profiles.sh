#!/bin/bash
# Everything profiles/ holds for one round, from one build: bench lines of every BASELINE.json configuration, rocprofv3
# kernel stats and PMC counters of the headline, level-20 and grade configurations, the single-GPU rehearsal of the
# decomposed step.  bash scripts/gpu_profiles.sh r02   (then copy gpurun_out/<tag>_profiles/* into profiles/)
TAG=${1:-r02}
OUT=gpurun_out/${TAG}_profiles
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout -k 10 500 python bench.py --steps 200 --warmup 20 > $OUT/${TAG}_bench.json 2> $OUT/bench.err; echo "bench rc=$?"
for wl in small2k wre20 grades; do
  timeout -k 10 400 python bench.py --workload $wl --steps 100 --warmup 10 --cpu-seconds 8 > $OUT/${TAG}_bench_$wl.json 2> $OUT/bench_$wl.err; echo "$wl rc=$?"
done
for wl in w16 small2k wre20 grades; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$wl -- python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-whole-step > $OUT/stats_$wl.json 2> $OUT/stats_$wl.err; echo "stats $wl rc=$?"
  find $OUT/stats_$wl -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $OUT/${TAG}_${wl}_kernel_stats.csv
done
[ -n "$SKIP_REHEARSAL" ] || { bash scripts/gpu_rehearsal.sh $TAG > $OUT/rehearsal.log 2>&1; echo "rehearsal rc=$?"; }
MTP_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29733 bench.py --gpus 2 --steps 20 --warmup 3 > $OUT/${TAG}_bench_2rank_gloo_rehearsal.json 2> $OUT/g2.err; echo "2-rank rehearsal rc=$?"
rm -rf $OUT/stats_*/
ls -la $OUT | head -40
