# round 4, call J: deferred reduction walking the partial slabs in 16-byte pieces (ReduceDesc mode 2) - suite, reduce_multi time, step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/${1:-r04j}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; grep -E "passed|failed|error" $O/gpu_suite.log | tail -3
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-alt --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('darcy b64', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --batch 256 --no-cpu-baseline --no-roofline --no-alt --steps 15 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('darcy b256', d['value'], d['ms_per_step'], d['step_flop_fraction'])"
done 2>&1 | tee $O/step.txt
timeout 300 python bench.py --workload mechanics --no-cpu-baseline --no-roofline --no-alt --steps 10 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mechanics', d['value'], d['ms_per_step'])"
for b in 64 256; do
(cd /tmp && PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$b -o p -- python $R/bench.py --batch $b --steps 8 --warmup 4 --no-cpu-baseline --no-alt --no-roofline > $O/prof$b.log 2>&1)
python - $O/prof$b $b <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/p_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
n=[int(r['Calls']) for r in rows if 'pack_multi' in r['Name']][0]
print('batch',sys.argv[2],'kernel ms/step', round(sum(int(r['TotalDurationNs']) for r in rows)/n/1e6,3))
for r in rows:
    if 'reduce_multi' in r['Name'] or 'pack_multi' in r['Name']: print('   ', r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
done
find $O -name '*.db' -delete; find $O -name '*kernel_trace.csv' -delete
