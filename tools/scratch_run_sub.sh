for k in ${KCAPS:-6 9}; do
  echo "== kcap $k stamps: $(python tools/ubench_subcycle.py 1024 $k 2>&1 | tail -3 | cut -c1-260)"
  for m in 0 1 2 4 8 15; do
    echo "== kcap $k skip $m: $(env IPD_BENCH_NODBG=1 IPD_DEBUG_SKIP=$m python tools/ubench_subcycle.py 1024 $k 2>&1 | tail -1 | cut -c1-45)"
  done
  echo "== kcap $k NO_BLKDENSE: $(env IPD_BENCH_NODBG=1 IPD_NO_BLKDENSE=1 python tools/ubench_subcycle.py 1024 $k 2>&1 | tail -1 | cut -c1-45)"
done
