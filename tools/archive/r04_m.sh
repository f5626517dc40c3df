# round 4, call M: row-streaming conv, one or two waves per SIMD (PIDM_CONV_RS_WPS) and rows per strip
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/${1:-r04m}; mkdir -p $O
SH="64,32,0,32,3,1,1,0;64,32,32,32,3,1,1,0;64,32,0,64,3,1,1,0"
for b in 64 256; do
  for v in "PIDM_CONV_RS=0" "PIDM_CONV_RS_WPS=1" "PIDM_CONV_RS_WPS=2" "PIDM_CONV_RS_WPS=2 PIDM_CONV_RS_WAVES=1024" "PIDM_CONV_RS_WPS=2 PIDM_CONV_RS_WAVES=4096 PIDM_CONV_RS_MINR=4"; do
    echo "#### batch $b  $v"
    env $v BENCH_CONV_SHAPES="$SH" timeout 300 python tools/bench_conv.py $b 2>&1 | grep -v "TOTAL\|amdgpu.ids" | cut -c1-110
  done
done > $O/shapes.txt 2>&1
for v in "PIDM_CONV_RS_WPS=1" "PIDM_CONV_RS_WPS=2"; do
  for b in 64 256; do
    echo "#### batch $b  $v"
    env $v timeout 600 python bench.py --batch $b --steps 30 --warmup 10 --no-cpu-baseline --no-alt --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done > $O/step.txt 2>&1
cat $O/shapes.txt $O/step.txt
