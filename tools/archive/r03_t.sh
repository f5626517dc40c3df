# split conv kernels with incremental stage cursors (no per-stage integer divisions): A/B against the previous build, same box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
C=physicsinformeddiffusionmodels_amd/csrc
timeout 600 python -m pytest tests/test_kernels_conv.py tests/test_unet_engine.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
cp $C/libpidm_hip.so $C/libpidm_hip_new.so
for v in new head new head; do
  cp $C/libpidm_hip_$v.so $C/libpidm_hip.so
  echo "== $v"; python tools/bench_conv.py 64 2>/dev/null | grep -E "K=3|k3|3x3|TOTAL" | cut -c1-130 | head -14
  timeout 300 python bench.py --no-cpu-baseline --no-alt --no-roofline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
done
cp $C/libpidm_hip_new.so $C/libpidm_hip.so
