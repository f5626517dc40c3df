# round 5, first GPU call: GPU suite -> A/B of the grouped weight gradients (batch 64 / 256) -> Darcy one-launch A/B -> the full bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r05a}; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log; grep -E "passed|failed|rc=|Error" $o/pytest.log | tail -5
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); l=d['launches']
print(f\"$1: {d['value']:9.1f} samples/s {d['ms_per_step']:8.3f} ms/step  kernels/step {l['kernels_inside_graphs_per_step']+l['kernels_enqueued_one_by_one_per_step']:.0f}\")"; }
for rep in 1 2; do
for b in 64 256; do
st=40; [ $b -ge 256 ] && st=15
for cfg in "PIDM_WGRAD_GROUP=0" "PIDM_WGRAD_GROUP=64" "PIDM_WGRAD_GROUP=64 PIDM_WGRAD_GROUP_SPLITDIV=2" "PIDM_WGRAD_GROUP=10" "PIDM_WGRAD_GROUP=64 PIDM_WGRAD_GROUP_SPLITDIV=4"; do
  env $cfg timeout 600 python bench.py --batch $b --no-cpu-baseline --no-alt --no-roofline --steps $st --warmup 8 2>>$o/bench.err | tail -1 | line "b$b $cfg"
done; done; done | tee $o/wgrad_group_ab.txt
for f in 1 0; do PIDM_DARCY_FUSED_FINALIZE=$f python tools/bench_darcy.py 2>&1 | sed "s/^/fused_finalize=$f /"; done | tee $o/darcy_ab.txt
timeout 900 python bench.py --steps 20 --warmup 5 2>>$o/bench.err | tail -1 > $o/bench.json
python - $o <<'PY'
import json,sys
d=json.load(open(sys.argv[1]+"/bench.json"))
r=d["roofline"]
print(d["value"], d["ms_per_step"], "sff", d["step_flop_fraction"], "launches", d["launches"]["kernels_inside_graphs_per_step"], d["launches"]["kernels_enqueued_one_by_one_per_step"])
for k in ("achieved","frac","frac_bf16_pipe","ceiling_split_form_tflops","step_frac_of_split_ceiling","conv_split_frac_of_split_ceiling","frac_at_measured_clock","all_kernels_ms_per_step","kernel_classes_ms_per_step"): print(k, r.get(k))
for t in r["top_kernels"]: print(t)
for k in ("fp32_mfma_only","eager_scalars","dropin_main_py","north_star_b256","mechanics_b32","sampling_b1024","residual_only"): print(k, d.get(k))
PY
tail -5 $o/bench.err
