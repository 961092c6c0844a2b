#!/bin/bash
# Round 3, final code: every file under profiles/r3_* in two gpurun calls (PART=1: the bench lines and the
# rocprofv3 passes of the default bench command; PART=2: stamps, remote tail, driver runs and their kernel
# statistics).  Run on the GPU box from the repo root; results in gpurun_out/ under the names profiles/ uses.
OUT=gpurun_out
if [ "${PART:-1}" = "1" ]; then
  bash tools/collect_profiles.sh r3 > $OUT/collect_profiles.log 2>&1
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/r3_bench_n1_steps20.json
  timeout -k 10 300 python3 bench.py --mask tree 2>/dev/null | tail -1 > $OUT/r3_bench_tree_v.json
  timeout -k 10 300 python3 bench.py --mask tree --cycle w 2>/dev/null | tail -1 > $OUT/r3_bench_tree_w.json
  timeout -k 10 300 python3 bench.py --mask newton --newton-k 9 --cycle w 2>/dev/null | tail -1 > $OUT/r3_bench_newton10_w.json
  timeout -k 10 300 python3 bench.py --mask newton --newton-k 30 --cycle w 2>/dev/null | tail -1 > $OUT/r3_bench_newton31_w.json
  echo part1 done
else
  bash tools/collect_r3_extras.sh > $OUT/collect_extras.log 2>&1
  bash tools/scratch_trace_step.sh > /dev/null 2>&1
  cp $OUT/trace_step.txt $OUT/r3_newton_step_trace.txt
  echo part2 done
fi
