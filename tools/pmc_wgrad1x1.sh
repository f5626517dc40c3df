# HBM reads and L2 hit / miss counts of the 1x1 weight-gradient launches of one batch-256 step (ungrouped at that batch):
# FETCH_SIZE (x2 on gfx950: 128-byte requests tallied at 64) and TCC_HIT / TCC_MISS in separate --pmc passes, kernel-trace only.
R=${GRAFT_REPO_ROOT:-/root/repo}; B=${1:-256}; o=$R/gpurun_out/r06_i/pmc_w1; mkdir -p $o
export TMPDIR=/tmp PYTHONPATH=$R
for v in "1 1" "0 0"; do
 set -- $v
 for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum"; do
  t=$(echo $c | cut -d' ' -f1)
  (cd /tmp && PIDM_WGRAD1X1_SPLIT=$1 PIDM_WGRAD1X1_XCD=$2 PIDM_GRAPH=0 timeout 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $o/$1$2_$t -o p -- python $R/tools/one_step.py $B 1 > $o/$1$2_$t.log 2>&1)
 done
done
python - $o <<'P'
import csv,sys,glob,os
o=sys.argv[1]
for tag in ('11','00'):
    res={}
    for d in sorted(glob.glob(o+'/'+tag+'_*')):
        if not os.path.isdir(d): continue
        f=glob.glob(d+'/*counter_collection.csv')
        if not f: print('no counters in',d); continue
        rows=list(csv.DictReader(open(f[0])))
        rows=[r for r in rows if 'conv_wgrad_1x1' in r['Kernel_Name']]
        # last step only: keep the second half of the dispatches
        ids=sorted({int(r['Dispatch_Id']) for r in rows}); half=ids[len(ids)//2:]
        for r in rows:
            if int(r['Dispatch_Id']) not in half: continue
            key=(half.index(int(r['Dispatch_Id'])), r['Kernel_Name'].split('(')[0][-30:], r['Grid_Size'])
            res.setdefault(key,{})[r['Counter_Name']]=float(r['Counter_Value'])
    print('split xcd =',tag)
    for k in sorted(res):
        c=res[k]
        fetch=c.get('FETCH_SIZE',0)*2*64/1e6 if 'FETCH_SIZE' in c else float('nan')   # 64-byte units, doubled on gfx950
        hit,miss=c.get('TCC_HIT_sum',0),c.get('TCC_MISS_sum',0)
        print('  %2d %-32s grid %-8s HBM read %8.1f MB  L2 hit %6.1f M miss %6.1f M (hit rate %.2f) req %6.1f M'%(k[0],k[1],k[2],fetch,hit/1e6,miss/1e6,hit/max(hit+miss,1),c.get('TCC_REQ_sum',0)/1e6))
P
