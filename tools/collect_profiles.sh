#!/bin/bash
# Collects the evidence bench.py's roofline object cites (run on the GPU box from the repo root; results land in
# gpurun_out/ under the names profiles/ uses -- copy them over afterwards).  PART=1: rocprofv3 --kernel-trace
# --stats and separate --pmc FETCH_SIZE / WRITE_SIZE passes of the default bench command, of `--n1 2048` and of
# one `--mask newton` line (PMC summaries keyed by workload: tools/summarize_pmc.py), then the bench lines
# themselves.  PART=2: stamps, realistic W cycles, driver runs and their kernel statistics, one Newton step traced.
set -o pipefail
R=${R:-r4}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
pmc_pass() {   # name, workload string, bench arguments...
  local name=$1 wl=$2; shift 2
  cd /tmp && export TMPDIR=/tmp
  rm -rf $OUT/prof_fetch_$name $OUT/prof_write_$name
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch_$name -o pf -- python3 $ROOT/bench.py --no-cpu-baseline --steps 50 --warmup 5 "$@" > $OUT/prof_fetch_$name.log 2>&1 || return 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write_$name -o pw -- python3 $ROOT/bench.py --no-cpu-baseline --steps 50 --warmup 5 "$@" > $OUT/prof_write_$name.log 2>&1 || return 1
  cd $ROOT
  local F=$(find $OUT/prof_fetch_$name -name "*counter_collection.csv" | head -1)
  local W=$(find $OUT/prof_write_$name -name "*counter_collection.csv" | head -1)
  python3 tools/summarize_pmc.py $OUT/${R}_pmc_summary_$name.json FETCH_SIZE=$F WRITE_SIZE=$W RESIDENT_CYCLES=5,50 WORKLOAD=$wl || return 1
  rm -rf $OUT/prof_fetch_$name $OUT/prof_write_$name
  echo "pmc $name done"
}
if [ "${PART:-1}" = "1" ]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $OUT/prof_stats
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o st -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/prof_stats.log 2>&1 || exit 1
  cd $ROOT
  S=$(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1)
  T=$(find $OUT/prof_stats -name "*kernel_trace.csv" | head -1)
  cp $S $OUT/${R}_kernel_stats.csv
  # the dispatches of the dominant kernel, one row each (the 200-cycle one is the timed region)
  python3 - "$T" > $OUT/${R}_resident_dispatches.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("kernel,start_ns,end_ns,duration_us,grid,workgroup,vgpr,lds_bytes")
for r in rows:
    if "k_resident" in r["Kernel_Name"]:
        print("%s,%s,%s,%.3f,%s,%s,%s,%s" % (r["Kernel_Name"].split("(")[0], r["Start_Timestamp"], r["End_Timestamp"],
              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")),
              r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("VGPR_Count", ""), r.get("LDS_Block_Size", "")))
PY
  rm -rf $OUT/prof_stats
  echo stats done
  pmc_pass n1 n1:1024,mask:bernoulli,rho:1,cycle:v || exit 1
  pmc_pass n2048 n1:2048,mask:bernoulli,rho:1,cycle:v --n1 2048 || exit 1
  pmc_pass newton31w n1:1024,mask:newton,cycle:w,newton_k:30 --mask newton --newton-k 30 --cycle w || exit 1
  pmc_pass newton2048w n1:2048,mask:newton,cycle:w,newton_k:24 --mask newton --n1 2048 --newton-k 24 --cycle w || exit 1
  timeout -k 10 400 python3 bench.py 2>$OUT/${R}_bench_n1.err | tail -1 > $OUT/${R}_bench_n1.json || exit 1
  timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${R}_bench_n1_steps20.json
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-poly2 2>/dev/null | tail -1 > $OUT/${R}_bench_n1_sweeps.json
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --cycle w 2>/dev/null | tail -1 > $OUT/${R}_bench_n1_w.json
  IPD_NO_RESIDENT=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${R}_bench_n1_multilaunch.json
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --n1 2048 2>/dev/null | tail -1 > $OUT/${R}_bench_n2048.json
  timeout -k 10 300 python3 bench.py --mask tree 2>/dev/null | tail -1 > $OUT/${R}_bench_tree_v.json
  timeout -k 10 300 python3 bench.py --mask tree --cycle w 2>/dev/null | tail -1 > $OUT/${R}_bench_tree_w.json
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --mask hub --cycle w 2>/dev/null | tail -1 > $OUT/${R}_bench_hub_w.json
  timeout -k 10 300 python3 bench.py --mask newton --newton-k 9 --cycle w 2>/dev/null | tail -1 > $OUT/${R}_bench_newton10_w.json
  timeout -k 10 300 python3 bench.py --mask newton --newton-k 30 --cycle w 2>/dev/null | tail -1 > $OUT/${R}_bench_newton31_w.json
  timeout -k 10 300 python3 bench.py --mask newton --n1 2048 --newton-k 24 --cycle w 2>/dev/null | tail -1 > $OUT/${R}_bench_newton25_n2048_w.json
  echo part1 done
else
  timeout -k 10 300 python3 tools/resident_stamps.py > $OUT/${R}_resident_stamps.txt 2>&1
  timeout -k 10 300 python3 tools/resident_stamps.py --no-poly2 >> $OUT/${R}_resident_stamps.txt 2>&1
  timeout -k 10 300 python3 tools/resident_stamps.py --cycle w >> $OUT/${R}_resident_stamps.txt 2>&1
  rm -rf /tmp/dump
  STRIDE=12 COUNT=12 timeout -k 10 400 python tools/bench_remote_tail.py 60-224 > $OUT/${R}_remote_tail.txt 2>&1
  IPD_PROFILE=1 timeout -k 10 600 python tools/bench_driver.py --sizes 500,1024,2048,4096 --classes 1,2 > $OUT/${R}_driver_runs.txt 2>&1
  bash tools/prof_driver.sh 1024 > $OUT/${R}_driver_kernel_stats.txt 2>&1
  bash tools/prof_driver.sh 2048 > $OUT/${R}_driver_kernel_stats_n2048.txt 2>&1
  SIZE=1024 WHICH=60 bash tools/trace_newton_step.sh > /dev/null 2>&1
  cp $OUT/trace_step.txt $OUT/${R}_newton_step_trace.txt
  SIZE=2048 WHICH=150 bash tools/trace_newton_step.sh > /dev/null 2>&1
  cp $OUT/trace_step.txt $OUT/${R}_newton_step_trace_n2048.txt
  timeout -k 10 300 python tools/bench_kkt.py > $OUT/${R}_kkt_wall.txt 2>&1
  bash tools/prof_kkt.sh > $OUT/${R}_kkt_kernel_stats.txt 2>&1
  echo part2 done
fi
