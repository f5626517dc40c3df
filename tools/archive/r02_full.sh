R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r02f}; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -q -x > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>$o/bench.err | tail -1 > $o/bench.json
python bench.py --workload mechanics --steps 10 --warmup 3 --no-cpu-baseline 2>>$o/bench.err | tail -1 > $o/bench_mechanics.json
python bench.py --workload sampling --steps 20 --warmup 5 --no-cpu-baseline 2>>$o/bench.err | tail -1 > $o/bench_sampling.json
tail -3 $o/pytest.log; for f in bench bench_mechanics bench_sampling; do python - $o/$f.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d.get("roofline") or {}
print(d["metric"][:40], d["value"], d["ms_per_step"], "alt", (d.get("fp32_mfma_only") or {}).get("ms_per_step"), "conv", r.get("achieved"), r.get("frac"), "fwd", r.get("fwd_dgrad"), "wgrad", r.get("wgrad"))
PY
done
