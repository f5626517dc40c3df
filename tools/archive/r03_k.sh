R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r03k}; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log; tail -3 $o/pytest.log
PIDM_BENCH_FORCE_EXCHANGE=1 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-alt --no-roofline 2>$o/force.err | tail -1 > $o/force_exchange.json
python - $o/force_exchange.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("forced exchange:", d["value"], d["ms_per_step"], d["exchange"], d["launches"]["graph_launches_per_step"])
PY
timeout 600 python bench.py --workload mechanics --steps 10 --warmup 4 --no-cpu-baseline --no-roofline 2>$o/mech.err | tail -1 > $o/bench_mechanics.json
python - $o/bench_mechanics.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("mechanics:", d["value"], d["ms_per_step"], {k:(d.get(k) or {}).get("value") for k in ("fp32_mfma_only","eager_scalars","dropin_main_py")})
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
