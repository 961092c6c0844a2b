#!/bin/bash
# The -m gpu suite once per A/B switch (every fallback path is kept green); one line per switch.
# PART=a|b runs one half (a whole sweep does not fit one 20-minute gpurun call)
ALL_A="IPD_NO_BPOLY IPD_NO_BLKDENSE IPD_NO_BLK IPD_NO_POLY IPD_NO_MIS_SMALL IPD_NO_SUBCYCLE"
ALL_B="IPD_NO_RESIDENT_REMOTE IPD_NO_RESIDENT_THREE IPD_NO_PAD IPD_NO_DONOR IPD_NO_STEP_DONOR IPD_NO_RESIDENT IPD_NO_RESIDENT_BIG IPD_NO_RESIDENT_DEEP IPD_NO_RES_POLY4"
case "${PART:-ab}" in a) LIST="$ALL_A";; b) LIST="$ALL_B";; c) LIST="$SWITCHES";; *) LIST="$ALL_A $ALL_B";; esac   # PART=c: SWITCHES="..."
OUT=gpurun_out/r4_switches_${PART:-ab}.txt
: > $OUT
for sw in $LIST; do
  env $sw=1 timeout -k 10 400 python -m pytest tests -m gpu -q > /tmp/sw.log 2>&1
  res=$(tail -1 /tmp/sw.log)
  fails=$(grep "^FAILED" /tmp/sw.log | cut -c1-150 | tr '\n' ';')
  echo "$sw=1: $res $fails" >> $OUT
  echo "$sw done"
done
