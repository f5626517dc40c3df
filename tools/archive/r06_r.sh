R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r06r}; rm -rf $o; mkdir -p $o
for d in 1 2 4; do
PIDM_LAP_NPER_DIV=$d timeout 600 python -m pytest tests/test_kernels_attn_proj.py tests/test_unet_engine.py -m gpu -x -q 2>&1 | tail -1
for b in 64 256; do
PIDM_LAP_NPER_DIV=$d timeout 600 python bench.py --batch $b --no-cpu-baseline --no-alt --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('div=$d batch $b', d['value'], d['ms_per_step'])" | tee -a $o/step_ab.txt
done
PIDM_LAP_NPER_DIV=$d timeout 600 python bench.py --workload sampling --no-cpu-baseline --no-alt --no-roofline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('div=$d sampling', d['value'], d['ms_per_step'])" | tee -a $o/step_ab.txt
done
