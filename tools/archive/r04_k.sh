# round 4, call K: block budget of the GroupNorm kernels (PIDM_GN_BLOCKS)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/${1:-r04k}; mkdir -p $O
for v in 1024 2048 4096 512; do
  for b in 64 256; do
  (cd /tmp && PIDM_GN_BLOCKS=$v PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${v}_$b -o p -- python $R/bench.py --batch $b --steps 8 --warmup 4 --no-cpu-baseline --no-alt --no-roofline > $O/prof_${v}_$b.log 2>&1)
  python - $O/prof_${v}_$b $v $b <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/p_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
n=[int(r['Calls']) for r in rows if 'pack_multi' in r['Name']][0]
out=[]
for r in rows:
    if 'gn_' in r['Name']: out.append(f"{r['Name'].split('(')[0].replace('pidm::','')}: {int(r['TotalDurationNs'])/n/1e3:.0f} us")
print('blocks',sys.argv[2],'batch',sys.argv[3],'total',round(sum(int(r['TotalDurationNs']) for r in rows)/n/1e6,3),' | '.join(out))
PY
  done
done
find $O -name '*.db' -delete; find $O -name '*kernel_trace.csv' -delete
