R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r03g}; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log; tail -3 $o/pytest.log
for r in 16 8 4; do PIDM_DARCY_ROWS=$r python tools/bench_darcy.py 2>/dev/null | tee -a $o/darcy.txt; done
(cd /tmp && PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_darcy -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $o/prof_darcy.log 2>&1)
find $o -name '*.db' -delete; find $o -name '*agent_info.csv' -delete; find $o -name '*kernel_trace.csv' -delete
python - $o/prof_darcy <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r['TotalDurationNs']) for r in rows)/1e6
print(f"kernel time total {tot:.1f} ms over the run (30 steps) = {tot/30:.3f} ms/step")
for r in rows:
    n=r['Name']
    if any(k in n for k in ("conv7x7","igemm_kernel<8","darcy","gn_","smallc")):
        print(f"{n.replace('void pidm::','').replace('pidm::','')[:60]:60s} calls={r['Calls']:>6} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={r['Percentage']}")
PY
tail -1 $o/prof_darcy.log | cut -c1-300
