R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r06e}; rm -rf $o; mkdir -p $o
export BENCH_CONV_SHAPES="16,128,0,384,1,1,0,0;16,128,0,128,1,1,0,0;8,256,0,768,1,1,0,0;16,128,0,128,3,1,1,0;16,256,0,128,3,1,1,0;8,256,0,256,3,1,1,0;8,512,0,256,3,1,1,0;32,64,0,64,3,1,1,0;32,128,0,64,3,1,1,0;16,64,0,128,4,2,1,0"
for b in 64 256; do for x in 0 1; do
echo "#### batch $b PIDM_SPLIT_XCD=$x" >> $o/conv_shapes.txt
PIDM_SPLIT_XCD=$x timeout 300 python tools/bench_conv.py $b 2>/dev/null | cut -c1-112 >> $o/conv_shapes.txt
done; done
cat $o/conv_shapes.txt
for rep in 1 2; do for x in 0 1; do for b in 64 256; do
PIDM_SPLIT_XCD=$x timeout 600 python bench.py --batch $b --no-cpu-baseline --no-alt --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('xcd=$x batch $b', d['value'], d['ms_per_step'])" | tee -a $o/step_ab.txt
done; done; done
for sh in "8 256 256 64"; do for x in 0 1; do
echo "######## PIDM_SPLIT_XCD=$x shape $sh" >> $o/trace.txt
PIDM_SPLIT_XCD=$x timeout 120 python tools/conv_trace.py $sh 2>&1 | head -8 >> $o/trace.txt
done; done
cat $o/trace.txt
