# LDS row pad of the split conv kernels at the 8- / 16-wide levels: A/B (PIDM_SPLIT_ROWPAD=0/1), per-shape us, PMC bank conflicts, bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
timeout 600 python -m pytest tests/test_kernels_conv.py tests/test_unet_engine.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
for p in 0 1 0 1; do echo "== PIDM_SPLIT_ROWPAD=$p"; PIDM_SPLIT_ROWPAD=$p python tools/bench_conv.py 64 2>/dev/null | grep -E "^ *(16|8) " | cut -c1-120; done
for p in 0 1 0 1; do PIDM_SPLIT_ROWPAD=$p timeout 300 python bench.py --no-cpu-baseline --no-alt --no-roofline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rowpad=$p', d['value'], d['ms_per_step'])"; done
for p in 0 1; do
  PIDM_SPLIT_ROWPAD=$p bash tools/pmc.sh sp16_$p $R/tools/bench_one.py 16 128 0 128 3 1 1 0 64 3 > /dev/null 2>&1
  PIDM_SPLIT_ROWPAD=$p bash tools/pmc.sh sp8_$p $R/tools/bench_one.py 8 256 0 256 3 1 1 0 64 3 > /dev/null 2>&1
  for n in sp16_$p sp8_$p; do echo "#### $n"; python tools/pmc_report.py gpurun_out/pmc_$n conv3x3_split | grep -E "==|waves="; done
done
find gpurun_out -name "*.db" -delete
