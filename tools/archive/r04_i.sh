# round 4, call I: pixels per block of la_bwd_pix_mfma_kernel (PIDM_LA_PPB) and the scalar kernel at the 8x8 level
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/${1:-r04i}; mkdir -p $O
for v in 0 128 64 32; do
  [ $v = 0 ] && unset PIDM_LA_PPB || export PIDM_LA_PPB=$v
  (cd /tmp && PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o p -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-alt --no-roofline > $O/prof_$v.log 2>&1)
  python - $O/prof_$v $v <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/p_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'la_bwd_pix' in r['Name']: print('ppb', sys.argv[2], r['Name'][:40], r['Calls'], 'avg', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
PY
done
find $O -name '*.db' -delete; find $O -name '*kernel_trace.csv' -delete
