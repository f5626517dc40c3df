# GPU pass: full -m gpu suite + conv ablation probe + per-shape conv table of the default bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r02c}; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
timeout 600 python tools/conv_probe.py > $o/conv_probe.txt 2>&1
PIDM_PROF_DUMP=$o/shape_table.txt timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $o/bench_short.json 2>$o/bench_short.err
tail -6 $o/pytest.log; cat $o/conv_probe.txt | tail -30
