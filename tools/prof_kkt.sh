#!/bin/bash
# rocprofv3 kernel statistics of tools/bench_kkt.py (ASAt / Ax / Aty at m=n=1024); run on the GPU box from
# the repo root.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/prof_kkt3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kkt3 -o k -- python3 $ROOT/tools/bench_kkt.py --reps 50 > $OUT/prof_kkt3.log 2>&1
S=$(find $OUT/prof_kkt3 -name "*kernel_stats.csv" | head -1)
python3 - "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print("%-60s calls %6s avg %8.2f us min %8.2f max %8.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
