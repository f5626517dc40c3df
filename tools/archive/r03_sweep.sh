# full training step vs per-GPU batch (one box, one call): samples/s, ms/step, FLOP fraction of the step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; mkdir -p gpurun_out
for b in 16 32 64 128 256 512; do
st=30; [ $b -ge 256 ] && st=10
timeout 600 python bench.py --batch $b --no-cpu-baseline --no-alt --no-roofline --steps $st --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
b=$b
print(f'batch {b:4d}: {d[\"value\"]:8.1f} samples/s  {d[\"ms_per_step\"]:8.3f} ms/step  step_flop_fraction {b*11.916e9/(d[\"ms_per_step\"]*1e-3)/157.3e12:.3f}')"
done | tee gpurun_out/batch_sweep.txt
