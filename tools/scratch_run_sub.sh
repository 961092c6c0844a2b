for k in 30 20; do
  for m in 0 4 8; do
    echo "== kcap $k skip $m: $(env IPD_BENCH_NODBG=1 IPD_DEBUG_SKIP=$m python tools/ubench_subcycle.py 1024 $k 2>&1 | tail -1 | cut -c1-45)"
  done
  echo "== kcap $k NO_LMAP: $(env IPD_BENCH_NODBG=1 IPD_NO_LMAP=1 python tools/ubench_subcycle.py 1024 $k 2>&1 | tail -1 | cut -c1-45)"
done
