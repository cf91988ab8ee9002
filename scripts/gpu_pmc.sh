#!/bin/bash
# PMC counter passes for the dominant kernel (separate rocprofv3 runs, --kernel-trace only, as the guide prescribes)
# plus a --stats pass; writes $OUT/pmc_counters.json (per-launch means, tagged with the kernel source hash) and
# $OUT/kernel_stats.csv -- copy both into profiles/ to have bench.py quote them.
#   bash scripts/gpu_pmc.sh TAG [bench.py args, e.g. --workload wre20]
OUT=gpurun_out/${1:-pmc}; shift
ARGS="$@"
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
run() {
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline $ARGS > $OUT/$name.json 2> $OUT/$name.err
  echo "$name rc=$?"
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
run tcc3 TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum
run mfma SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline $ARGS > $OUT/stats.json 2> $OUT/stats.err; echo "stats rc=$?"
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $OUT/kernel_stats.csv
head -8 $OUT/kernel_stats.csv
python - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, ".")
from lammps_mtp_kokkos_amd import capi
args = "$ARGS".split()
workload = args[args.index("--workload") + 1] if "--workload" in args else "w16"
tot, others = {}, collections.defaultdict(dict)
for name in ["sq1","sq2","sq3","tcc1","tcc2","tcc3","mfma"]:
    files = glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            kn = row.get("Kernel_Name","")
            if "mtp_wave_kernel" in kn:
                acc[("", row["Counter_Name"])].append(float(row["Counter_Value"]))
            else:
                for short in ("mtp_grade_kernel_os", "mtp_grade_kernel_lds", "mtp_grade_kernel", "mtp_cvec_kernel", "mtp_ev_finish"):
                    if short in kn:
                        acc[(short, row["Counter_Name"])].append(float(row["Counter_Value"]))
                        break
    for (kn, k), v in acc.items():
        if kn == "":
            print("%s %-28s mean per launch %.6g  (n=%d)" % (name, k, sum(v)/len(v), len(v)))
            tot[k] = sum(v)/len(v)
        else:
            others[kn][k] = sum(v)/len(v)
out = {"source_hash": capi.kernel_source_hash(), "workload": workload, "cells": 32, "kernel": "mtp_wave_kernel",
       "counters": tot, "other_kernels": others,
       "note": "rocprofv3 --kernel-trace --pmc, separate passes, mean per launch of mtp_wave_kernel under bench.py. "
               "FETCH_SIZE / WRITE_SIZE in KB as reported: FETCH_SIZE is taken as is (the guide's x2 correction is calibrated "
               "for 16 B/lane streams, these reads are 4-8 B gathers: uncalibrated), WRITE_SIZE is exact for atomics."}
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    out["hbm_bytes_per_launch"] = (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024
    out["fetch_doubled_bound_bytes"] = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024
json.dump(out, open("$OUT/pmc_counters.json", "w"), indent=1)
PY
