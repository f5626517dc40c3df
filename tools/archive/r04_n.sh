# round 4, call N: GPU suite at the row-streaming conv build + the kernel's clock / cycles per row (PIDM_RS_TRACE) + whole step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/${1:-r04n}; mkdir -p $O
STEP_VARIANTS=${STEP_VARIANTS-"PIDM_CONV_RS=0 PIDM_CONV_RS=1"}
if [ -z "$SKIP_TESTS" ]; then timeout 1500 python -m pytest tests/ -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log; fi
for b in 64 256; do
  echo "#### batch $b"
  PIDM_RS_TRACE=1 BENCH_CONV_SHAPES="64,32,0,32,3,1,1,0;64,32,32,32,3,1,1,0;64,32,0,64,3,1,1,0;32,64,0,64,3,1,1,0" python tools/bench_conv.py $b 2>&1 | grep "row loop" | python -c "
import sys, re, collections
d = collections.OrderedDict()
for l in sys.stdin:
    m = re.search(r'(conv3x3_rs_kernel<[^>]*> R=\\d+): row loop ([\\d.]+) us, shader clock ([\\d.]+) GHz, (\\d+) cycles', l)
    if m: d.setdefault(m.group(1), []).append(tuple(float(x) for x in m.group(2, 3, 4)))
for k, v in d.items():
    n = len(v); print('%-40s launches %3d  row loop %6.1f us  shader clock %.3f GHz  %5.0f cycles per input row' % (k, n, sum(x[0] for x in v) / n, sum(x[1] for x in v) / n, sum(x[2] for x in v) / n))
"
done > $O/clock.txt 2>&1
for v in $STEP_VARIANTS; do
  for b in 64 256; do
    echo "#### batch $b  $v"
    env $v timeout 600 python bench.py --batch $b --steps 30 --warmup 10 --no-cpu-baseline --no-alt --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done > $O/step.txt 2>&1
cat $O/clock.txt $O/step.txt
