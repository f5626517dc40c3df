# round 4, call E: 4x4 / stride-2 weight gradients on the row-streaming bf16 kernel (conv_wgrad_rs4_kernel) - GPU parity, per-shape and step A/B
# against the fp32-MFMA conv_wgrad_pipe_kernel<2,2,..> (PIDM_WGRAD_RS=0 turns both row-streaming kernels off: the 3x3 shapes are listed for reference)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/${1:-r04e}; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_conv.py tests/test_unet_engine.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
SH="64,32,0,32,4,2,1,0;32,64,0,64,4,2,1,0;16,128,0,128,4,2,1,0;8,256,0,256,4,2,1,1;16,128,0,128,4,2,1,1;32,64,0,64,4,2,1,1"
for b in 64 256; do
  for v in 1 0 1 0; do
    echo "== 4x4/s2 shapes, batch $b, PIDM_WGRAD_RS=$v"
    BENCH_CONV_SHAPES="$SH" PIDM_WGRAD_RS=$v timeout 300 python tools/bench_conv.py $b 2>&1 | grep -E "^H=|TOTAL" | sed 's/| fwd.*| wgrad/| wgrad/'
  done
done > $O/wgrad4_ab.txt 2>&1
cat $O/wgrad4_ab.txt
for v in 1 0 1 0; do
  PIDM_WGRAD_RS=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-alt --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('darcy b64 rs=$v', d['value'], d['ms_per_step'])"
  PIDM_WGRAD_RS=$v timeout 300 python bench.py --batch 256 --no-cpu-baseline --no-roofline --no-alt --steps 15 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('darcy b256 rs=$v', d['value'], d['ms_per_step'], d['step_flop_fraction'])"
done 2>&1 | tee $O/step_ab.txt
