# round 5, second GPU call: Darcy one-launch (agent-scope atomics) A/B + its tests; split divisor of the grouped weight gradients at batch 16 / 64 / 256
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r05b}; rm -rf $o; mkdir -p $o
timeout 600 python -m pytest tests/test_kernels_darcy.py tests/test_gpu_fullsize.py -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log; grep -E "passed|failed|rc=|Error" $o/pytest.log | tail -5
for rep in 1 2; do for f in 1 0; do PIDM_DARCY_FUSED_FINALIZE=$f python tools/bench_darcy.py 2>&1 | grep -v amdgpu.ids | sed "s/^/fused_finalize=$f /"; done; done | tee $o/darcy_ab.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); l=d['launches']
print(f\"$1: {d['value']:9.1f} samples/s {d['ms_per_step']:8.3f} ms/step  kernels/step {l['kernels_inside_graphs_per_step']+l['kernels_enqueued_one_by_one_per_step']:.0f}\")"; }
for rep in 1 2; do
for b in 16 64 256; do
st=40; [ $b -ge 256 ] && st=15
for cfg in "PIDM_WGRAD_GROUP=0" "PIDM_WGRAD_GROUP_SPLITDIV=2" "PIDM_WGRAD_GROUP_SPLITDIV=4" "PIDM_WGRAD_GROUP_SPLITDIV=8" "PIDM_WGRAD_GROUP_SPLITDIV=16" "PIDM_WGRAD_GROUP_SPLITDIV=32"; do
  env $cfg timeout 600 python bench.py --batch $b --no-cpu-baseline --no-alt --no-roofline --steps $st --warmup 8 2>>$o/bench.err | tail -1 | line "b$b $cfg"
done; done; done | tee $o/wgrad_splitdiv.txt
