R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r06l}; rm -rf $o; mkdir -p $o
for rep in 1 2 3; do for cfg in "PIDM_LIBRARY=$R/tools/ab/libpidm_hip_r05.so" "PIDM_X=0"; do for b in 16 64 256; do
env $cfg timeout 600 python bench.py --batch $b --no-cpu-baseline --no-alt --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg batch $b', d['value'], d['ms_per_step'])" | tee -a $o/step_ab.txt
done; done; done
