R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r06i}; rm -rf $o; mkdir -p $o
export PIDM_SPLIT_MS=1
timeout 900 python -m pytest tests/test_kernels_conv.py -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log; grep -E "passed|failed|rc=" $o/pytest.log | tail -3
for sh in "8 256 256 64"; do for cfg in "PIDM_X=0" "PIDM_SPLIT_NPW=4"; do
echo "######## $cfg shape $sh" >> $o/trace.txt
env $cfg timeout 120 python tools/conv_trace.py $sh 2>&1 | head -9 >> $o/trace.txt
done; done
cat $o/trace.txt
for rep in 1 2; do for cfg in "PIDM_X=0" "PIDM_SPLIT_NPW=4"; do for b in 64 256; do
env $cfg timeout 600 python bench.py --batch $b --no-cpu-baseline --no-alt --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg batch $b', d['value'], d['ms_per_step'])" | tee -a $o/step_ab.txt
done; done; done
