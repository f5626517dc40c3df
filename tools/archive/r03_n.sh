R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
timeout 300 python -m pytest tests/test_kernels_conv.py -m gpu -x -q -k "accurate or bit_identical" -s 2>&1 | grep -E "max error|passed|failed"
echo "== chains=1"; python tools/bench_conv.py 64 2>/dev/null | grep -E "K=3|TOTAL" | cut -c1-110
timeout 300 python bench.py --no-cpu-baseline --no-alt --no-roofline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('chains=1', d['value'], d['ms_per_step'])"
C=physicsinformeddiffusionmodels_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DPIDM_SPLIT_CHAINS=0 -x hip -c $C/k_conv.hip -o $C/build/k_conv.hip.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libpidm_hip.so $C/build/*.o
echo "== chains=0"; python tools/bench_conv.py 64 2>/dev/null | grep -E "K=3|TOTAL" | cut -c1-110
timeout 300 python bench.py --no-cpu-baseline --no-alt --no-roofline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('chains=0', d['value'], d['ms_per_step'])"
timeout 300 python -m pytest tests/test_kernels_conv.py -m gpu -x -q -k "accurate" -s 2>&1 | grep -E "max error|passed|failed"
