#!/bin/bash
# LDS counters of the dominant kernel for two settings of one environment knob (one rocprofv3 --pmc pass each):
#   bash scripts/gpu_pmc_lds.sh TAG "ENV=a" "ENV=b" [-- bench.py args]
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
CFGS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do CFGS+=("$1"); shift; done; [ "$1" == "--" ] && shift
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
i=0
for cfg in "${CFGS[@]}"; do
  i=$((i+1))
  [ "$cfg" != "-" ] && export $cfg
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/c$i -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-whole-step "$@" > $OUT/c$i.json 2> $OUT/c$i.err
  echo "[$cfg] rc=$?"
  [ "$cfg" != "-" ] && unset ${cfg%%=*}
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/c$i/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "mtp_wave_kernel" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("   " + "  ".join("%s %.4g" % (k, sum(v) / len(v)) for k, v in sorted(acc.items())))
PY
done
