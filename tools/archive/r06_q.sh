R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r06q}; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_unet_engine.py tests/test_training_step.py tests/test_gpu_fullsize.py tests/test_guidance.py tests/test_cocogen_correction.py -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log; grep -E "passed|failed|rc=" $o/pytest.log | tail -3
for rep in 1 2 3; do for cfg in "PIDM_NO_GN_INPLACE=1" "PIDM_X=0"; do
env $cfg timeout 600 python bench.py --workload sampling --no-cpu-baseline --no-alt --no-roofline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg sampling', d['value'], d['ms_per_step'])" | tee -a $o/step_ab.txt
done; done
