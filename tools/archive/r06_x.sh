R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r06x}; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_kernels_attn_proj.py tests/test_unet_engine.py tests/test_training_step.py tests/test_gpu_fullsize.py -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log; grep -E "passed|failed|rc=" $o/pytest.log | tail -3
for rep in 1 2 3; do for cfg in "PIDM_X=0" "PIDM_LAP_MIN_N=2048"; do for b in 64 256; do
env $cfg timeout 600 python bench.py --batch $b --no-cpu-baseline --no-alt --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg batch $b', d['value'], d['ms_per_step'])" | tee -a $o/step_ab.txt
done; done; done
