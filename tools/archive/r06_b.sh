# round 6: A/B of the producer waves' hand-counted vmcnt (PIDM_WS_LEAVE_FETCH): conv kernel tests, per-shape conv times, step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r06b}; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_kernels_conv.py tests/test_unet_engine.py tests/test_gpu_fullsize.py -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log; grep -E "passed|failed|rc=" $o/pytest.log | tail -3
export BENCH_CONV_SHAPES="16,128,0,384,1,1,0,0;16,128,0,128,1,1,0,0;8,256,0,768,1,1,0,0;8,128,0,256,1,1,0,0;16,128,0,128,3,1,1,0;8,256,0,256,3,1,1,0;8,512,0,128,3,1,1,0;32,64,0,64,3,1,1,0;16,64,0,128,4,2,1,0;8,256,0,128,4,2,1,1"
for b in 64 256; do
for lib in base new; do
echo "#### batch $b lib $lib" >> $o/conv_shapes.txt
L=""; [ $lib = base ] && L=$R/tools/ab/libpidm_hip_base.so
PIDM_LIBRARY=$L timeout 300 python tools/bench_conv.py $b >> $o/conv_shapes.txt 2>/dev/null
echo "#### batch $b lib $lib PIDM_SPLIT_WS=1" >> $o/conv_shapes.txt
PIDM_SPLIT_WS=1 PIDM_LIBRARY=$L timeout 300 python tools/bench_conv.py $b >> $o/conv_shapes.txt 2>/dev/null
done; done
cat $o/conv_shapes.txt | cut -c1-125
for rep in 1 2; do for lib in base new; do
L=""; [ $lib = base ] && L=$R/tools/ab/libpidm_hip_base.so
for b in 64 256; do
PIDM_LIBRARY=$L timeout 600 python bench.py --batch $b --no-cpu-baseline --no-alt --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib batch $b', d['value'], d['ms_per_step'])" | tee -a $o/step_ab.txt
done; done; done
PIDM_SPLIT_WS=1 timeout 600 python bench.py --no-cpu-baseline --no-alt --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('new ws=1 batch 64', d['value'], d['ms_per_step'])" | tee -a $o/step_ab.txt
