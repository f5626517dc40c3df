# linear-attention tile groups (PIDM_LAP_GROUPS): GPU parity + timing; conv knob sweeps per shape
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
timeout 600 python -m pytest tests/test_kernels_attn_proj.py tests/test_kernels_conv.py -m gpu -x -q 2>&1 | tail -2
for hd in 4 2 1 8; do
python tools/bench_lap.py 64 64 $hd 32; PIDM_LAP_GROUPS=1 python tools/bench_lap.py 64 64 $hd 32
done
python tools/bench_lap.py 64 32 4 64; PIDM_LAP_GROUPS=1 python tools/bench_lap.py 64 32 4 64
run() { env "$1" timeout 300 python bench.py $2 --no-cpu-baseline --no-alt --no-roofline --steps $3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['ms_per_step'])"; }
run X=1 "" 40
run PIDM_WGRAD_SPLIT_P=256 "" 40
run X=1 "" 40
run PIDM_WGRAD_SPLIT_P=256 "" 40
for k in X=1 PIDM_STREAM_WGS=512 PIDM_STREAM_WGS=128 PIDM_SPLIT_NW=4 PIDM_SPLIT_NW=8 PIDM_SPLIT_ROWPAD=0 PIDM_SPLIT_ROWPAD=16; do
echo "== $k"; env $k python tools/bench_conv.py 64 2>/dev/null | grep -E "K=3" | cut -c1-130
done
