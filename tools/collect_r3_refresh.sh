#!/bin/bash
# Re-measures what the last change of round 3 moved (the realistic W cycles and the driver runs).
OUT=gpurun_out
timeout -k 10 300 python3 bench.py --mask newton --newton-k 9 --cycle w 2>/dev/null | tail -1 > $OUT/r3_bench_newton10_w.json
timeout -k 10 300 python3 bench.py --mask newton --newton-k 30 --cycle w 2>/dev/null | tail -1 > $OUT/r3_bench_newton31_w.json
timeout -k 10 300 python3 bench.py --mask tree 2>/dev/null | tail -1 > $OUT/r3_bench_tree_v.json
rm -rf /tmp/dump
STRIDE=12 COUNT=12 python tools/bench_remote_tail.py 60-224 > $OUT/r3_remote_tail.txt 2>&1
IPD_PROFILE=1 python tools/bench_driver.py --sizes 500,1024,2048,4096 --classes 1,2 > $OUT/r3_driver_runs.txt 2>&1
bash tools/scratch_prof_driver.sh > $OUT/r3_driver_kernel_stats.txt 2>&1
bash tools/scratch_trace_step.sh > /dev/null 2>&1
cp $OUT/trace_step.txt $OUT/r3_newton_step_trace.txt
echo refreshed
