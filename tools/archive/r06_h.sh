R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r06h}; rm -rf $o; mkdir -p $o
export PIDM_SPLIT_MS=1
for rep in 1 2; do for lib in "" nopk_all nopk_conv nopk_rs nopk_attn; do for b in 64 256; do
L=""; [ -n "$lib" ] && L=$R/tools/ab/libpidm_hip_$lib.so
PIDM_LIBRARY=$L timeout 600 python bench.py --batch $b --no-cpu-baseline --no-alt --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lib=$lib batch $b', d['value'], d['ms_per_step'])" | tee -a $o/step_ab.txt
done; done; done
