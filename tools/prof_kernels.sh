# rocprofv3 medians of the kernels whose name matches $1 (regex) in a batch-$2 Darcy step, new library and - if ab_old/ exists - the old one
pat=$1; b=${2:-64}; mkdir -p gpurun_out/r06_i; o=$PWD/gpurun_out/r06_i; R=$PWD
cd /tmp && export TMPDIR=/tmp
for lib in new old; do
  unset PIDM_LIBRARY; [ $lib = old ] && { [ -f $R/ab_old/libpidm_hip.so ] || continue; export PIDM_LIBRARY=$R/ab_old/libpidm_hip.so; }
  PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/pk_$lib -o p -- python $R/bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-alt --no-roofline > $o/pk_$lib.log 2>&1
  echo "$lib library, batch $b"; python - $o/pk_$lib/p_kernel_trace.csv "$pat" <<'P'
import csv,sys,statistics,re
rows=list(csv.DictReader(open(sys.argv[1]))); pat=re.compile(sys.argv[2])
d={}
for r in rows:
    n=r['Kernel_Name']
    if pat.search(n):
        k=n.split('(')[0].replace('void pidm::','').replace('pidm::','')
        d.setdefault(k,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
tot=0
for k,v in sorted(d.items()):
    print('   %-70s calls/step %5.1f  median %6.1f us  sum/step %7.1f us'%(k[:70],len(v)/23,statistics.median(v),sum(v)/23)); tot+=sum(v)/23
print('   total per step %.1f us'%tot)
P
done
