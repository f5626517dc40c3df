# kernel statistics of the 1x1 weight-gradient launches: split form with its own split plan for ungrouped launches (default) /
# the stream plan (PIDM_WGRAD1X1_SPLIT_PLAN=0) / fp32 streams (PIDM_WGRAD1X1_SPLIT=0), batch 64 and 256
mkdir -p gpurun_out/r06_i; o=$PWD/gpurun_out/r06_i; R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in "1 1" "1 0" "0 0"; do
 set -- $v
 for b in 64 256; do
  PIDM_WGRAD1X1_SPLIT=$1 PIDM_WGRAD1X1_SPLIT_PLAN=$2 PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/pw_$1$2_$b -o p -- python $R/bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-roofline > $o/pw_$1$2_$b.log 2>&1
  echo "split=$1 own plan=$2 batch $b"; python - $o/pw_$1$2_$b/p_kernel_trace.csv <<'P'
import csv,sys,statistics
rows=list(csv.DictReader(open(sys.argv[1])))
d={}
for r in rows:
    n=r['Kernel_Name']
    if 'conv_wgrad_1x1' in n or 'reduce_multi' in n:
        k=n.split('(')[0].replace('void pidm::','').replace('pidm::','')
        d.setdefault(k,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items():
    print('   %-45s calls/step %5.1f  median %7.1f us  sum/step %8.1f us'%(k,len(v)/13,statistics.median(v),statistics.median(v)*len(v)/13))
P
 done
done
