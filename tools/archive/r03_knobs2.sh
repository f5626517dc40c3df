# 3x3 weight gradient: 128- vs 256-pixel tile (PIDM_WGRAD_SPLIT_P), batch 64 / 256 / mechanics, one box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
run() { env "$1" timeout 300 python bench.py $2 --no-cpu-baseline --no-alt --no-roofline --steps $3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['ms_per_step'])"; }
for r in 1 2; do
run X=1 "" 40
run PIDM_WGRAD_SPLIT_P=128 "" 40
done
run X=1 "--batch 256" 10
run PIDM_WGRAD_SPLIT_P=128 "--batch 256" 10
run X=1 "--batch 256" 10
run PIDM_WGRAD_SPLIT_P=128 "--batch 256" 10
run X=1 "--workload mechanics" 20
run PIDM_WGRAD_SPLIT_P=128 "--workload mechanics" 20
run X=1 "--workload mechanics" 20
run PIDM_WGRAD_SPLIT_P=128 "--workload mechanics" 20
python tools/bench_conv.py 64 2>/dev/null | grep -E "K=3" | cut -c1-160
echo "== P=128"; PIDM_WGRAD_SPLIT_P=128 python tools/bench_conv.py 64 2>/dev/null | grep -E "K=3" | cut -c1-160
