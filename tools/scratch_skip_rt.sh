rm -rf /tmp/dump
for m in ${MODES:-0 1 2 4 8 15}; do
  echo "== IPD_DEBUG_SKIP=$m"
  IPD_DEBUG_SKIP=$m STRIDE=24 COUNT=3 python tools/bench_remote_tail.py 60-224 2>&1 | grep -E "^s|REMOTE=0" | cut -c1-75,100-330
done
