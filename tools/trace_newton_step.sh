#!/bin/bash
# kernel sequence of ONE Newton step of the m=n=1024 Class 1 run (between two resident launches): run on the
# GPU box from the repo root; WHICH=<index of the resident launch> (default 60), SIZE=<m=n> (default 1024);
# result in gpurun_out/trace_step.txt
ROOT=$(pwd); OUT=$ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/prof_trace
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_trace -o d -- python3 $ROOT/tools/bench_driver.py --sizes ${SIZE:-1024} --classes 1 > $OUT/prof_trace.log 2>&1
S=$(find $OUT/prof_trace -name "*kernel_trace.csv" | head -1)
python3 - "$S" "${WHICH:-60}" <<'PY' > $OUT/trace_step.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void k_resident")]
w = int(sys.argv[2])
a, b = idx[w], idx[w + 1]
t0 = int(rows[a]["End_Timestamp"])
print("step between resident launches %d and %d: %d kernels, %.1f us" % (w, w + 1, b - a - 1, (int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
prev = t0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f  gap %6.1f  dur %6.1f  %s  grid %s wg %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:60], r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?")))
    prev = e
PY
rm -rf $OUT/prof_trace
