#!/bin/bash
# Collects the evidence bench.py's roofline object cites (run on the GPU box from the repo root):
#   rocprofv3 --kernel-trace --stats, and separate --pmc FETCH_SIZE / WRITE_SIZE passes of the
#   default bench command; then the bench lines themselves.  Results land in gpurun_out/ with the
#   names profiles/ uses (copy them over afterwards).
set -o pipefail
R=${1:-r1}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o st -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/prof_stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o pf -- python3 $ROOT/bench.py --no-cpu-baseline --steps 50 --warmup 5 > $OUT/prof_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o pw -- python3 $ROOT/bench.py --no-cpu-baseline --steps 50 --warmup 5 > $OUT/prof_write.log 2>&1 || exit 1
cd $ROOT
F=$(find $OUT/prof_fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/prof_write -name "*counter_collection.csv" | head -1)
S=$(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1)
echo "fetch=$F write=$W stats=$S"
python3 tools/summarize_pmc.py $OUT/${R}_pmc_summary.json FETCH_SIZE=$F WRITE_SIZE=$W || exit 1
cp $OUT/${R}_pmc_summary.json profiles/${R}_pmc_summary.json
cp $S $OUT/${R}_kernel_stats.csv
timeout -k 10 400 python3 bench.py 2>/dev/null | tail -1 > $OUT/${R}_bench_n1.json || exit 1
timeout -k 10 300 python3 bench.py --no-cpu-baseline --maskop 2>/dev/null | tail -1 > $OUT/${R}_bench_n1_maskop.json
timeout -k 10 300 python3 bench.py --no-cpu-baseline --mask tree 2>/dev/null | tail -1 > $OUT/${R}_bench_tree_v.json
timeout -k 10 300 python3 bench.py --no-cpu-baseline --mask tree --cycle w 2>/dev/null | tail -1 > $OUT/${R}_bench_tree_w.json
timeout -k 10 300 python3 bench.py --no-cpu-baseline --mask hub --cycle w 2>/dev/null | tail -1 > $OUT/${R}_bench_hub_w.json
echo collected
