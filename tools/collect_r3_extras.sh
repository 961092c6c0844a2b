#!/bin/bash
# Round-3 evidence beyond collect_profiles.sh (run on the GPU box from the repo root; results in gpurun_out/).
OUT=gpurun_out
{
  echo "# tools/ubench_subcycle.py: one launch of the single-workgroup sub-cycle kernel (one W leg rooted at k_sub) on Newton"
  echo "# systems captured from the m=n=1024 Class 1 driver run at APD iterations 31 / 21 / 10; 'us per launch' without stamps"
  echo "# (IPD_BENCH_NODBG=1) is the figure to quote, the per-stage figures (with stamps) give proportions; then the same launch"
  echo "# with parts switched off (IPD_DEBUG_SKIP: 1 polynomial passes, 2 coarsest PCG, 4 row walks of the thread-per-row"
  echo "# sweeps, 8 those sweeps altogether, 15 all of it) and with the round's changes switched off one by one."
  for k in 30 20 9; do
    echo "== APD iteration $((k+1))"
    IPD_DEBUG_SWEEP=1 python tools/ubench_subcycle.py 1024 $k 2>&1 | tail -3
    for m in 0 1 2 4 8 15; do
      echo "   skip $m: $(IPD_BENCH_NODBG=1 IPD_DEBUG_SKIP=$m python tools/ubench_subcycle.py 1024 $k 2>&1 | tail -1 | cut -c1-44)"
    done
    echo "   IPD_NO_BPOLY=1:           $(IPD_BENCH_NODBG=1 IPD_NO_BPOLY=1 python tools/ubench_subcycle.py 1024 $k 2>&1 | tail -1 | cut -c1-44)"
    echo "   IPD_NO_BPOLY=1 IPD_NO_LPOLY=1 (first half of round 3): $(IPD_BENCH_NODBG=1 IPD_NO_BPOLY=1 IPD_NO_LPOLY=1 IPD_NO_BLKDENSE=1 python tools/ubench_subcycle.py 1024 $k 2>&1 | tail -1 | cut -c1-44)"
    echo "   IPD_NO_POLY=1:            $(IPD_BENCH_NODBG=1 IPD_NO_POLY=1 python tools/ubench_subcycle.py 1024 $k 2>&1 | tail -1 | cut -c1-44)"
    echo "   IPD_NO_LMAP=1:            $(IPD_BENCH_NODBG=1 IPD_NO_LMAP=1 python tools/ubench_subcycle.py 1024 $k 2>&1 | tail -1 | cut -c1-44)"
    echo "   IPD_NO_POLY=1 IPD_NO_LMAP=1 (round 2's sweeps): $(IPD_BENCH_NODBG=1 IPD_NO_POLY=1 IPD_NO_LMAP=1 python tools/ubench_subcycle.py 1024 $k 2>&1 | tail -1 | cut -c1-44)"
  done
} > $OUT/r3_subcycle_stamps.txt 2>&1
rm -rf /tmp/dump
STRIDE=12 COUNT=12 python tools/bench_remote_tail.py 60-224 > $OUT/r3_remote_tail.txt 2>&1
python tools/resident_stamps.py > $OUT/r3_resident_stamps.txt 2>&1
python tools/resident_stamps.py --cycle w >> $OUT/r3_resident_stamps.txt 2>&1
IPD_PROFILE=1 python tools/bench_driver.py --sizes 500,1024,2048,4096 --classes 1,2 > $OUT/r3_driver_runs.txt 2>&1
bash tools/scratch_prof_driver.sh > $OUT/r3_driver_kernel_stats.txt 2>&1
echo done
