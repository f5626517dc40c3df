# round-3 call A: GPU tests (incl. the single-rank RCCL exchange), the default bench line with its new legs, the forced-exchange line,
# a rocprofv3 kernel-stats baseline of the Darcy step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r03a}; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log; tail -5 $o/pytest.log
timeout 600 python bench.py 2>$o/bench.err | tail -1 > $o/bench.json; cut -c1-400 $o/bench.json
PIDM_BENCH_FORCE_EXCHANGE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-roofline 2>$o/force.err | tail -1 > $o/force_exchange.json
python - $o <<'PY'
import json,sys
o=sys.argv[1]
for f in ("bench.json","force_exchange.json"):
    try:
        d=json.load(open(f"{o}/{f}"))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, d["value"], d["ms_per_step"], {k:(d[k] if k in ("exchange","residual_only") else (d[k] or {}).get("value")) for k in ("fp32_mfma_only","eager_scalars","dropin_main_py","north_star_b256","residual_only","exchange")})
    if d.get("roofline"): print({k:v for k,v in d["roofline"].items() if k in ("achieved","frac","frac_bf16_pipe","split_form","fp32_mfma_form","kernel_ms_per_step","step_flop_fraction")})
PY
(cd /tmp && PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_darcy -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $o/prof_darcy.log 2>&1)
find $o -name '*.db' -delete; find $o -name '*agent_info.csv' -delete; find $o -name '*kernel_trace.csv' -delete
python - $o/prof_darcy <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True)
if f:
    rows=list(csv.DictReader(open(f[0]))); N=25+5   # warmup + steps + roofline steps
    tot=sum(float(r['TotalDurationNs']) for r in rows)/1e6
    print(f"kernel time total {tot:.1f} ms over the run")
    for r in rows[:28]:
        print(f"{r['Name'].replace('void pidm::','').replace('pidm::','')[:60]:60s} calls={r['Calls']:>6} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={r['Percentage']}")
PY
ls $o
