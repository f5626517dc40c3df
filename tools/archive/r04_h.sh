# round 4, call H: per-pixel LA backward on the matrix cores for the 8x8 level too and with fewer waves per block at batch 64 - suite + step + kernel times
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/${1:-r04h}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; grep -E "passed|failed|error" $O/gpu_suite.log | tail -3
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-alt --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('darcy b64', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --batch 256 --no-cpu-baseline --no-roofline --no-alt --steps 15 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('darcy b256', d['value'], d['ms_per_step'], d['step_flop_fraction'])"
done 2>&1 | tee $O/step.txt
(cd /tmp && PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $O/prof.log 2>&1)
python - $O <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/prof/**/p_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'la_' in r['Name'] or 'mid_attn' in r['Name']: print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
find $O -name '*.db' -delete; find $O -name '*kernel_trace.csv' -delete
