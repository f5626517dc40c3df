R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
timeout 300 python -m pytest tests/test_kernels_darcy.py tests/test_gpu_fullsize.py tests/test_cocogen_correction.py -m gpu -x -q 2>&1 | tail -2
for r in 8 12 16 28; do PIDM_DARCY_ROWS=$r python tools/bench_darcy.py 2>/dev/null | sed "s/^/quad /"; done
