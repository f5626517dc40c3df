# knob A/Bs on the final build (one box): each line = bench.py --steps 40 with one environment knob
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-alt --no-roofline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
run X=1
run PIDM_SPLIT_NW=4
run PIDM_WGRAD_SPLIT_P=128
run PIDM_NO_BN_EPILOGUE=1
run PIDM_NO_GN_EPILOGUE=1
run X=1
run PIDM_SPLIT_WS=1
run PIDM_SPLIT_WS=0
run PIDM_DARCY_ROWS=16
run PIDM_GRAPH=0
run X=1
