#!/bin/bash
# A/B of tuning knobs on the headline workload, each configuration timed ROUNDS times in alternation (box clocks drift):
#   bash scripts/gpu_ab.sh TAG "ENV1=a ENV2=b" "ENV1=c" ...      ("-" = defaults; VARIANTS="name:-DFLAG ..." builds
#   diagnostic libraries ab/libmtp_mi355x_<name>.so first, selectable with MTP_LIB=lammps_mtp_kokkos_amd/ab/libmtp_mi355x_<name>.so)
set -o pipefail
TAG=${1:-ab}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
for v in $VARIANTS; do
  make -s -C lammps_mtp_kokkos_amd/csrc variant NAME=${v%%:*} EXTRA="${v#*:}" >> $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
done
for r in $(seq 1 ${ROUNDS:-2}); do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    envs=""; [ "$cfg" != "-" ] && envs="$cfg"
    env $envs timeout -k 10 200 python bench.py --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline $BENCH_ARGS > $OUT/b${i}_$r.json 2> $OUT/b${i}_$r.err
    python - <<PY
import json
try:
    d=json.load(open("$OUT/b${i}_$r.json")); l=d["config"]["launch"]; print("r$r [$cfg] ms/step %.4f kernel_ms %.4f  wpb %d wps %d lds %d" % (d["ms_per_step"], d["roofline"]["kernel_ms"], l["waves_per_block"], l["waves_per_simd"], l["lds_bytes_per_wave"]))
except Exception as e:
    print("[$cfg] failed", e); print(open("$OUT/b${i}_$r.err").read()[-1500:])
PY
  done
done
