# round 4, call L: the row-streaming forward / dgrad kernel (k_conv_rs.hip) against conv3x3_split_kernel - unit tests, per-shape times, whole step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/${1:-r04l}; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_conv.py tests/test_unet_engine.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
SH="64,32,0,32,3,1,1,0;64,32,32,32,3,1,1,0;64,32,0,64,3,1,1,0;32,32,0,64,3,1,1,0;32,64,0,64,3,1,1,0;32,64,0,32,3,1,1,0"
for b in 64 256; do
  for v in "PIDM_CONV_RS=0" "PIDM_CONV_RS=1" "PIDM_CONV_RS=1 PIDM_CONV_RS_WAVES=2048" "PIDM_CONV_RS=1 PIDM_CONV_RS_WAVES=512"; do
    echo "#### batch $b  $v"
    env $v BENCH_CONV_SHAPES="$SH" timeout 300 python tools/bench_conv.py $b 2>&1 | grep -v "TOTAL\|amdgpu.ids" | cut -c1-110
  done
done > $O/shapes.txt 2>&1
for v in "PIDM_CONV_RS=0" "PIDM_CONV_RS=1"; do
  for b in 64 256; do
    echo "#### batch $b  $v"
    env $v timeout 600 python bench.py --batch $b --steps 30 --warmup 10 --no-cpu-baseline --no-alt --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done > $O/step.txt 2>&1
cat $O/shapes.txt $O/step.txt
