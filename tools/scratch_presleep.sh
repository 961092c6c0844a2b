rm -rf /tmp/dump
for p in ${PS:-0 6 10 13 16 20}; do
  echo "== IPD_RES_PRESLEEP=$p"
  IPD_RES_PRESLEEP=$p STRIDE=36 COUNT=4 python tools/bench_remote_tail.py 60-224 2>&1 | grep -E "REMOTE=0" | cut -c140-330
  IPD_RES_PRESLEEP=$p python tools/bench_driver.py --sizes 1024 --classes 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   driver', round(d['apd_s'],4))"
done
