#!/bin/bash
# rocprofv3 kernel statistics of one whole Class 1 driver run at m=n=1024 (run on the GPU box from the
# repo root; prints the 45 kernels with the largest total time).  Usage: bash tools/prof_driver.sh [SIZE]
ROOT=$(pwd); OUT=$ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/prof_drv3
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_drv3 -o d -- python3 $ROOT/tools/bench_driver.py --sizes ${1:-1024} --classes 1 > $OUT/prof_drv3.log 2>&1
S=$(find $OUT/prof_drv3 -name "*kernel_stats.csv" | head -1)
python3 - "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.3f s over %d kernels" % (tot / 1e9, len(rows)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print("%-70s calls %7s  total %8.2f ms  avg %8.2f us  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
