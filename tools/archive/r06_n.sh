R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r06n}; rm -rf $o; mkdir -p $o
for lib in "" prio; do
L=""; [ -n "$lib" ] && L=$R/tools/ab/libpidm_hip_$lib.so
for cfg in "PIDM_X=0" "PIDM_SPLIT_NPW=4"; do
echo "######## lib=$lib $cfg shape 8 256 256 64" >> $o/trace.txt
env $cfg PIDM_BENCH_LIB=$L timeout 120 python tools/conv_trace.py 8 256 256 64 2>&1 | head -9 | tail -6 >> $o/trace.txt
done; done
cat $o/trace.txt
for rep in 1 2; do for lib in "" prio; do for b in 64; do
L=""; [ -n "$lib" ] && L=$R/tools/ab/libpidm_hip_$lib.so
PIDM_LIBRARY=$L timeout 600 python bench.py --batch $b --no-cpu-baseline --no-alt --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lib=$lib batch $b', d['value'], d['ms_per_step'])" | tee -a $o/step_ab.txt
done; done; done
