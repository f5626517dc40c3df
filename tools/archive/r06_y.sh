#!/bin/bash
# round 6: GroupNorm block plan - at least 16 pixels per block (new) against 64 (until now), step time at batch 16 / 64 / 256,
# mechanics and sampling; alternating legs on one box
for rep in 1 2 3; do
  for mp in 64 16; do
    for b in 64; do
      PIDM_GN_MINPIX=$mp python bench.py --batch $b --steps 30 --warmup 8 --no-cpu-baseline --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('minpix $mp batch $b:', d['ms_per_step'], 'ms')"
    done
  done
done
for mp in 64 16; do
  PIDM_GN_MINPIX=$mp python bench.py --batch 16 --steps 30 --warmup 8 --no-cpu-baseline --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('minpix $mp batch 16:', d['ms_per_step'], 'ms')"
  PIDM_GN_MINPIX=$mp python bench.py --batch 256 --steps 10 --warmup 4 --no-cpu-baseline --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('minpix $mp batch 256:', d['ms_per_step'], 'ms')"
  PIDM_GN_MINPIX=$mp python bench.py --workload mechanics --steps 10 --warmup 4 --no-cpu-baseline --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('minpix $mp mechanics:', d['ms_per_step'], 'ms')"
  PIDM_GN_MINPIX=$mp python bench.py --workload sampling --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('minpix $mp sampling:', d['ms_per_step'], 'ms')"
done
