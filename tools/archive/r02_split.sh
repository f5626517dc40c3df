R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r02s}; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_kernels_conv.py tests/test_unet_engine.py tests/test_training_step.py -m gpu -q -x > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
PIDM_CONV_SPLIT=0 timeout 300 python tools/bench_conv.py 64 2>/dev/null | grep "K=3" | cut -c42-96 > $o/old.txt
timeout 300 python tools/bench_conv.py 64 2>/dev/null | grep "K=3" | cut -c1-96 > $o/new.txt
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>$o/bench.err | tail -1 > $o/bench.json
tail -3 $o/pytest.log; paste $o/new.txt $o/old.txt; cut -c1-260 $o/bench.json
