# lap_bwd per-kernel time, new build vs previous build (same box)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/${1:-r03p}; mkdir -p $O
C=physicsinformeddiffusionmodels_amd/csrc
cp $C/libpidm_hip.so $C/libpidm_hip_new.so
for v in new head; do
  cp $C/libpidm_hip_$v.so $C/libpidm_hip.so
  (cd /tmp && PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-roofline > $O/prof_$v.log 2>&1)
  python - <<PY
import csv,glob
f=glob.glob('$O/prof_$v/**/p_kernel_stats.csv',recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    tot+=float(r['TotalDurationNs'])
    if 'lap_' in r['Name'] or 'layernorm' in r['Name']: print('$v', r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1))
print('$v total kernel ms', tot/1e6)
PY
done
cp $C/libpidm_hip_new.so $C/libpidm_hip.so
