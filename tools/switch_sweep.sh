#!/bin/bash
# The -m gpu suite once per A/B switch (every fallback path is kept green); one line per switch.
OUT=gpurun_out/r3_switches.txt
: > $OUT
for sw in IPD_NO_BLK IPD_NO_POLY IPD_NO_LMAP IPD_NO_MIS_SMALL IPD_NO_ASAT_SMALL IPD_NO_UPLOAD_RING IPD_NO_MAILBOX \
          IPD_NO_SUBCYCLE IPD_NO_SEMI IPD_NO_SEMI_ROOT IPD_NO_RESIDENT_REMOTE IPD_NO_RESIDENT_THREE IPD_RES_NO_XMASK \
          IPD_NO_TINY IPD_NO_PAD IPD_NO_FUSE IPD_NO_DONOR IPD_NO_STEP_DONOR IPD_NO_RESIDENT IPD_NO_RESIDENT_BIG IPD_GALERKIN_SMALL; do
  env $sw=1 timeout -k 10 400 python -m pytest tests -m gpu -q > /tmp/sw.log 2>&1
  res=$(tail -1 /tmp/sw.log)
  fails=$(grep "^FAILED" /tmp/sw.log | cut -c1-150 | tr '\n' ';')
  echo "$sw=1: $res $fails" >> $OUT
  echo "$sw done"
done
