#!/bin/bash
# round 6: GroupNorm prologues (independent loads requested before the statistics chain) - the build before against HEAD, one box,
# alternating legs; then the per-level kernel durations of HEAD (tools/gn_by_shape.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { env "$@" python bench.py --batch $B --steps 30 --warmup 8 --no-cpu-baseline --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$LBL batch $B:', d['ms_per_step'], 'ms')"; }
python -m pytest tests/test_unet_engine.py tests/test_training_step.py -x -q -m gpu 2>&1 | tail -1
for rep in 1 2 3; do
  for B in 64; do
    LBL=before run PIDM_LIBRARY=$R/tools/ab/libpidm_gn_before.so
    LBL=head run PIDM_GN_MINPIX=16
  done
done
B=16; LBL=before run PIDM_LIBRARY=$R/tools/ab/libpidm_gn_before.so; LBL=head run PIDM_GN_MINPIX=16
B=256; LBL=before run PIDM_LIBRARY=$R/tools/ab/libpidm_gn_before.so; LBL=head run PIDM_GN_MINPIX=16
for w in mechanics sampling; do
  for l in before head; do
    if [ $l = before ]; then export PIDM_LIBRARY=$R/tools/ab/libpidm_gn_before.so; else unset PIDM_LIBRARY; fi
    python bench.py --workload $w --steps 10 --warmup 4 --no-cpu-baseline --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$l $w:', d['ms_per_step'], 'ms')"
  done
done
unset PIDM_LIBRARY
bash tools/gn_by_shape.sh r06_gn64_after 64
