# 1x1 convolutions as a split-form GEMM (conv1x1_split_kernel): tests, A/B against PIDM_CONV1X1_SPLIT=0 (same build, same box), per-kernel times
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/${1:-r03w}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed"
for v in 1 0 1 0; do
  PIDM_CONV1X1_SPLIT=$v timeout 300 python bench.py --no-cpu-baseline --no-alt --no-roofline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('darcy 1x1split=$v', d['value'], d['ms_per_step'])"
done
for v in 1 0 1 0; do
  PIDM_CONV1X1_SPLIT=$v timeout 300 python bench.py --workload mechanics --no-cpu-baseline --no-alt --no-roofline --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mechanics 1x1split=$v', d['value'], d['ms_per_step'])"
done
for v in 1 0; do
  (cd /tmp && PIDM_CONV1X1_SPLIT=$v PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-roofline > $O/prof_$v.log 2>&1)
  python - <<PY
import csv,glob
f=glob.glob('$O/prof_$v/**/p_kernel_stats.csv',recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    tot+=float(r['TotalDurationNs'])
    if 'conv_igemm' in r['Name'] or 'conv1x1' in r['Name']: print('split=$v', r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,2), round(float(r['TotalDurationNs'])/25e6,3))
print('split=$v total kernel ms/step', tot/25e6)
PY
done
