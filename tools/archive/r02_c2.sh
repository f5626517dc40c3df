R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r02u}; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_kernels_conv.py tests/test_unet_engine.py tests/test_training_step.py tests/test_gpu_fullsize.py -m gpu -q -x > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
tail -3 $o/pytest.log
PIDM_CONV_SPLIT=0 timeout 300 python tools/bench_conv.py 64 2>/dev/null | grep -E "K=4" | cut -c1-140 > $o/old.txt
timeout 300 python tools/bench_conv.py 64 2>/dev/null | grep -E "K=4" | cut -c1-140 > $o/new.txt
echo "--- split"; cat $o/new.txt; echo "--- fp32"; cat $o/old.txt
for i in 1 2; do python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --no-alt 2>/dev/null | tail -1 | cut -c1-170; done
python bench.py --workload mechanics --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-alt 2>/dev/null | tail -1 | cut -c1-200
