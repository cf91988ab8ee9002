#!/bin/bash
# Experiment (round 2, led to the 12-wavefront plan and the SIMD-balance rule of plan()): does a third wavefront per SIMD pay?  A level-14 potential (smaller per-atom LDS image) lets 12 wavefronts
# per CU fit today; the 168-VGPR build is timed at 8 and at 12 wavefronts per CU, next to the shipped build.
OUT=gpurun_out/${1:-occ}
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
make -s -C lammps_mtp_kokkos_amd/csrc variant NAME=w3 EXTRA=-DMTP_WAVES_PER_SIMD=3 >> $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
python -m lammps_mtp_kokkos_amd.mtpgen --level 14 --out /tmp/W_L14.mtp > $OUT/gen.log 2>&1
python -m lammps_mtp_kokkos_amd.mtpgen --level 12 --out /tmp/W_L12.mtp >> $OUT/gen.log 2>&1
POTF=/tmp/W_L14.mtp
run() {
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --potential $POTF --steps 100 --warmup 10 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/$name.json")); print("$name: ms/step %.4f kernel_ms %.4f launch %s" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"]["launch"]))
except Exception as e:
    print("$name failed", e); print(open("$OUT/$name.err").read()[-800:])
PY
}
run base_8 A=1
run w3_8 MTP_LIB=$PWD/lammps_mtp_kokkos_amd/libmtp_mi355x_w3.so MTP_MAX_WAVES=8
run w3_12 MTP_LIB=$PWD/lammps_mtp_kokkos_amd/libmtp_mi355x_w3.so MTP_MAX_WAVES=12
run w3_12_rows0 MTP_LIB=$PWD/lammps_mtp_kokkos_amd/libmtp_mi355x_w3.so MTP_MAX_WAVES=12 MTP_ROWS_LDS=0
run w3_10 MTP_LIB=$PWD/lammps_mtp_kokkos_amd/libmtp_mi355x_w3.so MTP_MAX_WAVES=10
POTF=/tmp/W_L12.mtp
run L12_base_8 A=1
run L12_w3_8 MTP_LIB=$PWD/lammps_mtp_kokkos_amd/libmtp_mi355x_w3.so MTP_MAX_WAVES=8
run L12_w3_12 MTP_LIB=$PWD/lammps_mtp_kokkos_amd/libmtp_mi355x_w3.so MTP_MAX_WAVES=12
run L12_w3_12_wpb6 MTP_LIB=$PWD/lammps_mtp_kokkos_amd/libmtp_mi355x_w3.so MTP_MAX_WAVES=12 MTP_WPB=6
run L12_w3_12_wpb4 MTP_LIB=$PWD/lammps_mtp_kokkos_amd/libmtp_mi355x_w3.so MTP_MAX_WAVES=12 MTP_WPB=4
