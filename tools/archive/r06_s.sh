R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r06s}; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log; grep -E "passed|failed|rc=" $o/pytest.log | tail -3
for rep in 1 2 3; do for cfg in "PIDM_NO_LN_GN_SUMS=1" "PIDM_X=0"; do for b in 64 256; do
env $cfg timeout 600 python bench.py --batch $b --no-cpu-baseline --no-alt --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg batch $b', d['value'], d['ms_per_step'], d['launches']['kernels_inside_graphs_per_step'])" | tee -a $o/step_ab.txt
done; done; done
for cfg in "PIDM_NO_LN_GN_SUMS=1" "PIDM_X=0"; do
env $cfg timeout 600 python bench.py --workload mechanics --no-cpu-baseline --no-alt --no-roofline --steps 10 --warmup 4 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg mechanics', d['value'], d['ms_per_step'])" | tee -a $o/step_ab.txt
done
