R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r06f}; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_kernels_conv.py tests/test_unet_engine.py tests/test_gpu_fullsize.py -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log; grep -E "passed|failed|rc=" $o/pytest.log | tail -3
export BENCH_CONV_SHAPES="16,128,0,384,1,1,0,0;16,128,0,128,1,1,0,0;8,256,0,768,1,1,0,0;16,256,0,128,1,1,0,0;16,128,0,128,3,1,1,0;16,256,0,128,3,1,1,0;8,256,0,256,3,1,1,0;8,512,0,256,3,1,1,0;32,64,0,64,3,1,1,0;32,128,0,64,3,1,1,0;16,64,0,128,4,2,1,0;8,256,0,128,4,2,1,1"
for b in 64 256; do for cfg in "PIDM_SPLIT_MS=1" "PIDM_X=0" "PIDM_SPLIT_NPW=4"; do
echo "#### batch $b $cfg" >> $o/conv_shapes.txt
env $cfg timeout 300 python tools/bench_conv.py $b 2>/dev/null | cut -c1-112 >> $o/conv_shapes.txt
done; done
cat $o/conv_shapes.txt
for rep in 1 2; do for cfg in "PIDM_SPLIT_MS=1" "PIDM_X=0" "PIDM_SPLIT_NPW=4"; do for b in 64 256; do
env $cfg timeout 600 python bench.py --batch $b --no-cpu-baseline --no-alt --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg batch $b', d['value'], d['ms_per_step'])" | tee -a $o/step_ab.txt
done; done; done
for sh in "16 128 128 64" "16 128 128 256"; do for cfg in "PIDM_X=0" "PIDM_SPLIT_NPW=4"; do
echo "######## $cfg shape $sh" >> $o/trace.txt
env $cfg timeout 120 python tools/conv_trace.py $sh 2>&1 | head -9 >> $o/trace.txt
done; done
cat $o/trace.txt
