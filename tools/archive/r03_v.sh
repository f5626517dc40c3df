# weight re-pack kernel with 32-bit hierarchical index arithmetic and pair processing: A/B (mechanics and Darcy), same box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/${1:-r03v}; mkdir -p $O
C=physicsinformeddiffusionmodels_amd/csrc
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed"
cp $C/libpidm_hip.so $C/libpidm_hip_new.so
for v in new head new head; do
  cp $C/libpidm_hip_$v.so $C/libpidm_hip.so
  timeout 300 python bench.py --workload mechanics --no-cpu-baseline --no-alt --no-roofline --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v mechanics', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --no-cpu-baseline --no-alt --no-roofline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v darcy', d['value'], d['ms_per_step'])"
done
for v in new head; do
  cp $C/libpidm_hip_$v.so $C/libpidm_hip.so
  for w in mechanics darcy; do
  (cd /tmp && PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${v}_$w -o p -- python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-roofline > $O/prof_${v}_$w.log 2>&1)
  python - <<PY
import csv,glob
f=glob.glob('$O/prof_${v}_$w/**/p_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "pack_" in r["Name"] or "reduce_multi" in r["Name"]: print('$v $w', r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
  done
done
cp $C/libpidm_hip_new.so $C/libpidm_hip.so
