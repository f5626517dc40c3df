# GroupNorm kernels (no integer divisions, first pixels fetched before the statistics prologue) + wgrad tile cursor: A/B, same box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/${1:-r03u}; mkdir -p $O
C=physicsinformeddiffusionmodels_amd/csrc
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed"
cp $C/libpidm_hip.so $C/libpidm_hip_new.so
for v in new head new head; do
  cp $C/libpidm_hip_$v.so $C/libpidm_hip.so
  timeout 300 python bench.py --no-cpu-baseline --no-alt --no-roofline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
done
for v in new head; do
  cp $C/libpidm_hip_$v.so $C/libpidm_hip.so
  (cd /tmp && PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-roofline > $O/prof_$v.log 2>&1)
  python - <<PY
import csv,glob
f=glob.glob('$O/prof_$v/**/p_kernel_stats.csv',recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    tot+=float(r['TotalDurationNs'])
    if 'gn_' in r['Name'] or 'wgrad_split' in r['Name'] or 'conv3x3_split' in r['Name']: print('$v', r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,2), round(float(r['TotalDurationNs'])/25e6,3))
print('$v total kernel ms/step', tot/25e6)
PY
done
cp $C/libpidm_hip_new.so $C/libpidm_hip.so
