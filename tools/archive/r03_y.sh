# producer iterations with the DMA wait before the fetch (conv3x3 WS + 1x1 GEMM), producer-side epilogue of the 1x1 GEMM: A/B vs the previous build
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
C=physicsinformeddiffusionmodels_amd/csrc
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|AssertionError: \("
python tools/bench_conv1x1.py 64 2>&1 | tail -13
cp $C/libpidm_hip.so $C/libpidm_hip_new.so
for v in new head new head; do
  cp $C/libpidm_hip_$v.so $C/libpidm_hip.so
  timeout 300 python bench.py --no-cpu-baseline --no-alt --no-roofline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v darcy', d['value'], d['ms_per_step'])"
done
for v in new head; do
  cp $C/libpidm_hip_$v.so $C/libpidm_hip.so
  timeout 300 python bench.py --workload mechanics --no-cpu-baseline --no-alt --no-roofline --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v mechanics', d['value'], d['ms_per_step'])"
  echo "== $v conv (4-wave tile shapes)"; python tools/bench_conv.py 64 2>/dev/null | grep -E "H=  8|H= 16 Cin=  64" | cut -c1-110
done
cp $C/libpidm_hip_new.so $C/libpidm_hip.so
