#!/bin/bash
# Collects the evidence bench.py's roofline object cites (run on the GPU box from the repo root):
#   rocprofv3 --kernel-trace --stats, and separate --pmc FETCH_SIZE / WRITE_SIZE passes of the
#   default bench command; then the bench lines themselves.  Results land in gpurun_out/ with the
#   names profiles/ uses (copy them over afterwards).
set -o pipefail
R=${1:-r3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o st -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/prof_stats.log 2>&1 || exit 1
echo stats done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o pf -- python3 $ROOT/bench.py --no-cpu-baseline --steps 50 --warmup 5 > $OUT/prof_fetch.log 2>&1 || exit 1
echo fetch done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o pw -- python3 $ROOT/bench.py --no-cpu-baseline --steps 50 --warmup 5 > $OUT/prof_write.log 2>&1 || exit 1
echo write done
cd $ROOT
F=$(find $OUT/prof_fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/prof_write -name "*counter_collection.csv" | head -1)
S=$(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1)
T=$(find $OUT/prof_stats -name "*kernel_trace.csv" | head -1)
echo "fetch=$F write=$W stats=$S trace=$T"
python3 tools/summarize_pmc.py $OUT/${R}_pmc_summary.json FETCH_SIZE=$F WRITE_SIZE=$W RESIDENT_CYCLES=5,50 WORKLOAD=n1:1024,mask:bernoulli,rho:1,cycle:v || exit 1
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
timeout -k 10 400 python3 bench.py 2>$OUT/${R}_bench_n1.err | tail -1 > $OUT/${R}_bench_n1.json || exit 1
timeout -k 10 300 python3 bench.py --no-cpu-baseline --cycle w 2>/dev/null | tail -1 > $OUT/${R}_bench_n1_w.json
IPD_NO_RESIDENT=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${R}_bench_n1_multilaunch.json
timeout -k 10 300 python3 bench.py --no-cpu-baseline --n1 2048 2>/dev/null | tail -1 > $OUT/${R}_bench_n2048.json
timeout -k 10 300 python3 bench.py --no-cpu-baseline --mask tree 2>/dev/null | tail -1 > $OUT/${R}_bench_tree_v.json
timeout -k 10 300 python3 bench.py --no-cpu-baseline --mask tree --cycle w 2>/dev/null | tail -1 > $OUT/${R}_bench_tree_w.json
timeout -k 10 300 python3 bench.py --no-cpu-baseline --mask hub --cycle w 2>/dev/null | tail -1 > $OUT/${R}_bench_hub_w.json
timeout -k 10 300 python3 tools/resident_stamps.py > $OUT/${R}_resident_stamps.txt 2>&1
timeout -k 10 300 python3 tools/resident_stamps.py --cycle w >> $OUT/${R}_resident_stamps.txt 2>&1
echo collected
