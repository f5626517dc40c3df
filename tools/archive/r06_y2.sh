#!/bin/bash
# round 6: GroupNorm block plan - pixel floor 16 / 8 / 4 and block budget 1024 / 2048, batch 64 (and 16), alternating legs on one box
run() { env "$@" python bench.py --batch $B --steps 30 --warmup 8 --no-cpu-baseline --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$* batch $B:', d['ms_per_step'], 'ms')"; }
for rep in 1 2; do
  for B in 64 16; do
    run PIDM_GN_MINPIX=16
    run PIDM_GN_MINPIX=8
    run PIDM_GN_MINPIX=4
    run PIDM_GN_MINPIX=16 PIDM_GN_BLOCKS=2048
    run PIDM_GN_MINPIX=8 PIDM_GN_BLOCKS=2048
  done
done
