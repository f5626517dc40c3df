# GPU pass: streaming conv parity + A/B bench with per-shape tables
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r02f}; rm -rf $o; mkdir -p $o
timeout 600 python -m pytest tests/test_kernels_conv.py tests/test_unet_engine.py tests/test_training_step.py -m gpu -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
PIDM_CONV_STREAM=0 PIDM_PROF_DUMP=$o/shape_old.txt timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>$o/bench.err | tail -1 > $o/bench_old.json
PIDM_PROF_DUMP=$o/shape_new.txt timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>>$o/bench.err | tail -1 > $o/bench_new.json
timeout 300 python bench.py --steps 20 --warmup 5 --batch 256 --no-cpu-baseline --no-roofline 2>>$o/bench.err | tail -1 > $o/bench_new_b256.json
tail -3 $o/pytest.log; cut -c1-400 $o/bench_old.json; echo; cut -c1-400 $o/bench_new.json; echo; cut -c1-200 $o/bench_new_b256.json; echo
grep "k3x3" $o/shape_old.txt | grep "^0" | sort -k6 > /tmp/a.txt; grep "k3x3" $o/shape_new.txt | grep "^0" | sort -k6 > /tmp/b.txt; paste /tmp/a.txt /tmp/b.txt | awk '{printf "%-60s old %7.1fus %6.1fTF | new %7.1fus %6.1fTF\n", $6" "$7" "$8" "$9" "$10" "$13, $4, $5, $(NF/2+4), $(NF/2+5)}' | head -40
