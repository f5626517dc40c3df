# round 4, call A: row-streaming weight gradient (conv_wgrad_rs_kernel) - GPU parity, per-shape A/B against conv_wgrad_split_kernel
# (PIDM_WGRAD_RS=0, same build, same box), step-level A/B at batch 64 and 256
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/${1:-r04a}; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_conv.py -m gpu -x -q 2>&1 | tail -3
SH="64,32,0,32,3,1,1,0;64,32,32,32,3,1,1,0;32,64,0,64,3,1,1,0;32,64,64,32,3,1,1,0;16,128,0,128,3,1,1,0;16,128,128,64,3,1,1,0;8,256,0,256,3,1,1,0;8,256,256,128,3,1,1,0"
for b in 64 256; do
  for v in 1 0; do
    echo "== wgrad shapes, batch $b, PIDM_WGRAD_RS=$v"
    BENCH_CONV_SHAPES="$SH" PIDM_WGRAD_RS=$v timeout 300 python tools/bench_conv.py $b 2>&1 | tail -12
  done
done > $O/wgrad_ab.txt 2>&1
cat $O/wgrad_ab.txt
for v in 1 0 1 0; do
  PIDM_WGRAD_RS=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('darcy rs=$v', d['value'], d['ms_per_step'], 'b256', d['north_star_b256']['value'], d['north_star_b256']['ms_per_step'], d['north_star_b256']['step_flop_fraction'])"
done 2>&1 | tee $O/step_ab.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
