# round 5, third GPU call: GPU suite; grouped weight gradients by family at batch 16 / 64 / 256; mechanics + sampling sanity
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r05c}; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log; grep -E "passed|failed|rc=|Error" $o/pytest.log | tail -5
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); l=d['launches']
print(f\"$1: {d['value']:9.1f} {d['unit']} {d['ms_per_step']:8.3f} ms/step  kernels/step {l['kernels_inside_graphs_per_step']+l['kernels_enqueued_one_by_one_per_step']:.0f}\")"; }
for rep in 1 2; do
for b in 16 64 256; do
st=40; [ $b -ge 256 ] && st=15
for cfg in "PIDM_WGRAD_GROUP=0" "PIDM_WGRAD_GROUP_FAMS=1" "PIDM_WGRAD_GROUP_FAMS=3" "PIDM_WGRAD_GROUP_FAMS=5" "PIDM_WGRAD_GROUP_FAMS=7"; do
  env $cfg timeout 600 python bench.py --batch $b --no-cpu-baseline --no-alt --no-roofline --steps $st --warmup 8 2>>$o/bench.err | tail -1 | line "b$b $cfg"
done; done; done | tee $o/wgrad_group_fams.txt
timeout 600 python bench.py --workload mechanics --no-cpu-baseline --no-alt --no-roofline --steps 10 --warmup 4 2>>$o/bench.err | tail -1 | line "mechanics"
PIDM_GRAPH_BWD=1 timeout 600 python bench.py --workload mechanics --no-cpu-baseline --no-alt --no-roofline --steps 10 --warmup 4 2>>$o/bench.err | tail -1 | line "mechanics PIDM_GRAPH_BWD=1 (grouped)"
timeout 600 python bench.py --workload sampling --no-cpu-baseline --no-alt --no-roofline --steps 20 --warmup 5 2>>$o/bench.err | tail -1 | line "sampling"
