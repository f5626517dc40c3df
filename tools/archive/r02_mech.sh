R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r02o}; rm -rf $o; mkdir -p $o
timeout 600 python -m pytest tests/test_unet_engine.py tests/test_training_step.py -m gpu -q -k "dim128 or mech" > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
timeout 300 python bench.py --workload mechanics --steps 10 --warmup 3 --no-cpu-baseline 2>$o/bench.err | tail -1 > $o/bench_mech.json
timeout 300 python bench.py --workload sampling --steps 20 --warmup 5 --no-cpu-baseline 2>>$o/bench.err | tail -1 > $o/bench_samp.json
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>>$o/bench.err | tail -1 > $o/bench.json
tail -2 $o/pytest.log; for f in bench_mech bench_samp bench; do cut -c1-260 $o/$f.json; echo; done
