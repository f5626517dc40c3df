# round 4, call C: hand-laid-out iteration of the row-streaming weight gradient (PIDM_WGRAD_RS_VAR = 3 / 4) against variant 0, one box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/${1:-r04c}; mkdir -p $O
SH="64,32,0,32,3,1,1,0;64,32,32,32,3,1,1,0;32,64,0,64,3,1,1,0;16,128,0,128,3,1,1,0;8,256,0,256,3,1,1,0"
for b in 64 256; do
  for v in 0 3 4 0 3 4; do
    echo "== batch $b PIDM_WGRAD_RS_VAR=$v"
    BENCH_CONV_SHAPES="$SH" PIDM_WGRAD_RS_VAR=$v timeout 300 python tools/bench_conv.py $b 2>&1 | grep -E "^H=|TOTAL" | sed 's/| fwd.*| wgrad/| wgrad/'
  done
done > $O/wgrad_var.txt 2>&1
cat $O/wgrad_var.txt
for v in 3 4; do PIDM_WGRAD_RS_VAR=$v timeout 600 python -m pytest tests/test_kernels_conv.py -m gpu -x -q -k "split_forms or accurate or extreme" 2>&1 | grep -E "passed|failed"; done
for v in 0 3 0 3; do
  PIDM_WGRAD_RS_VAR=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('darcy var=$v', d['value'], d['ms_per_step'], 'b256', d['north_star_b256']['value'], d['north_star_b256']['ms_per_step'], d['north_star_b256']['step_flop_fraction'])"
done 2>&1 | tee $O/step_ab.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; grep -E "passed|failed|error" $O/gpu_suite.log | tail -3
