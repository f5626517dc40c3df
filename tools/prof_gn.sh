# rocprofv3 kernel statistics of the GroupNorm / LayerNorm launches of the Darcy step (batch 64 and 16), one line per kernel
mkdir -p gpurun_out/r06_i; o=$PWD/gpurun_out/r06_i; R=$PWD; tag=${1:-gn}
cd /tmp && export TMPDIR=/tmp
for b in 64 16; do
  PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_$b -o p -- python $R/bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-alt --no-roofline > $o/${tag}_$b.log 2>&1
  echo "batch $b: $(grep -o '"ms_per_step": [0-9.]*' $o/${tag}_$b.log | head -1)"; python - $o/${tag}_$b/p_kernel_trace.csv <<'P'
import csv,sys,statistics
rows=list(csv.DictReader(open(sys.argv[1])))
d={}
for r in rows:
    n=r['Kernel_Name']
    if 'gn_' in n or 'layernorm' in n:
        k=n.split('(')[0].replace('void pidm::','').replace('pidm::','')
        d.setdefault(k,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
tot=0
for k,v in sorted(d.items()):
    print('   %-34s calls/step %5.1f  median %6.1f us  mean %6.1f  sum/step %7.1f us'%(k,len(v)/23,statistics.median(v),statistics.mean(v),sum(v)/23)); tot+=sum(v)/23
print('   total per step %.1f us'%tot)
P
done
