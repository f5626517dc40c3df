R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r03f}; rm -rf $o; mkdir -p $o
timeout 300 python -m pytest tests/test_early_backward.py tests/test_training_step.py -m gpu -x -q 2>&1 | tail -2
for e in 1 0; do
PIDM_EARLY_BACKWARD=$e timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 30 2>$o/bench_early$e.err | tail -1 > $o/bench_early$e.json
python - $o/bench_early$e.json $e <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("early", sys.argv[2], d["value"], d["ms_per_step"], {k:(d.get(k) or {}).get("value") for k in ("fp32_mfma_only","eager_scalars","dropin_main_py","north_star_b256")})
PY
done
