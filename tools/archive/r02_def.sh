R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
timeout 600 python -m pytest tests/test_training_step.py -m gpu -q 2>&1 | tail -2
for e in "--eager-scalars" ""; do echo -n "scalars [$e]: "; python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --no-alt $e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; done
for e in "--eager-scalars" ""; do echo -n "mechanics scalars [$e]: "; python bench.py --workload mechanics --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-alt $e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; done
bash tools/r02_gaps.sh r02g 1 | tail -12
