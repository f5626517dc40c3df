# round 5, fourth GPU call: which problems to group - pixel threshold at batch 64 / 256 / 512
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r05d}; rm -rf $o; mkdir -p $o
timeout 600 python -m pytest tests/test_graph_replay.py tests/test_unet_engine.py -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log; grep -E "passed|failed|rc=|Error" $o/pytest.log | tail -5
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); l=d['launches']
print(f\"$1: {d['value']:9.1f} {d['unit']} {d['ms_per_step']:8.3f} ms/step  kernels/step {l['kernels_inside_graphs_per_step']+l['kernels_enqueued_one_by_one_per_step']:.0f}\")"; }
for rep in 1 2; do
for b in 64 256 512; do
st=40; [ $b -ge 256 ] && st=12
for cfg in "PIDM_WGRAD_GROUP=0" "PIDM_WGRAD_GROUP_MAXPIX=16384" "PIDM_WGRAD_GROUP_MAXPIX=65536" "PIDM_WGRAD_GROUP_MAXPIX=262144" "PIDM_WGRAD_GROUP_MAXPIX=1048576" "PIDM_WGRAD_GROUP_MAXPIX=99999999"; do
  env $cfg timeout 600 python bench.py --batch $b --no-cpu-baseline --no-alt --no-roofline --steps $st --warmup 8 2>>$o/bench.err | tail -1 | line "b$b $cfg"
done; done; done | tee $o/wgrad_group_maxpix.txt
