R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r03i}; rm -rf $o; mkdir -p $o
(cd /tmp && PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_b256 -o p -- python $R/bench.py --batch 256 --steps 6 --warmup 3 --no-cpu-baseline --no-alt > $o/prof_b256.log 2>&1)
find $o -name '*.db' -delete; find $o -name '*agent_info.csv' -delete; find $o -name '*kernel_trace.csv' -delete
tail -1 $o/prof_b256.log | cut -c1-200
python - $o/prof_b256 <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(f[0]))); N=6+3+5
tot=sum(float(r['TotalDurationNs']) for r in rows)/1e6
print(f"kernel time {tot/N:.3f} ms/step at batch 256")
for r in rows[:30]:
    print(f"{r['Name'].replace('void pidm::','').replace('pidm::','')[:60]:60s} calls/step={int(r['Calls'])/N:6.1f} avg_us={float(r['AverageNs'])/1e3:8.1f} ms/step={float(r['TotalDurationNs'])/1e6/N:7.3f}")
PY
