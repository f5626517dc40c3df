# round 4, call F: init-conv weight gradient on the row-streaming kernel (conv_wgrad_rs7_kernel): GPU parity, per-shape A/B, step A/B against the
# previous commit's library is not possible in one build - PIDM_WGRAD_RS=0 turns all three row-streaming kernels off (reference), the per-shape
# table isolates this one
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/${1:-r04f}; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_conv.py tests/test_unet_engine.py tests/test_training_step.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
SH="64,2,0,32,7,1,3,0;64,10,0,128,7,1,3,0;64,4,0,32,7,1,3,0"
for b in 64 256; do
  [ $b = 256 ] && SH="64,2,0,32,7,1,3,0;64,4,0,32,7,1,3,0"
  for v in 1 0 1 0; do
    echo "== 7x7 shapes, batch $b, PIDM_WGRAD_RS=$v"
    BENCH_CONV_SHAPES="$SH" PIDM_WGRAD_RS=$v timeout 300 python tools/bench_conv.py $b 2>&1 | grep -E "^H=" | sed 's/| fwd.*| wgrad/| wgrad/'
  done
done > $O/wgrad7_ab.txt 2>&1
cat $O/wgrad7_ab.txt
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-alt --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('darcy b64', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --batch 256 --no-cpu-baseline --no-roofline --no-alt --steps 15 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('darcy b256', d['value'], d['ms_per_step'], d['step_flop_fraction'])"
done 2>&1 | tee $O/step.txt
timeout 300 python bench.py --workload mechanics --no-cpu-baseline --no-roofline --no-alt --steps 10 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mechanics', d['value'], d['ms_per_step'])"
