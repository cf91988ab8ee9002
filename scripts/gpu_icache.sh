#!/bin/bash
# instruction-cache counters of the force kernel for two workloads (diagnostic)
OUT=gpurun_out/${1:-icache}
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_LEVEL|SQC_" | head -40 > $OUT/avail.txt
for wl in ${WLS:-small2k w16}; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_WAVE_CYCLES SQC_TC_INST_REQ --output-format csv -d $OUT/$wl -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-whole-step --workload $wl > $OUT/$wl.json 2> $OUT/$wl.err
  echo "$wl rc=$?"
done
python - <<PY
import csv, glob, collections
for wl in "${WLS:-small2k w16}".split():
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % wl, recursive=True):
        for row in csv.DictReader(open(f)):
            if "mtp_wave_kernel" in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(wl, {k: sum(v) / len(v) for k, v in acc.items()})
PY
head -30 $OUT/avail.txt
