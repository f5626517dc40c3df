# quick GPU pass: attention parity + bench + kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r02q}; rm -rf $o; mkdir -p $o
timeout 600 python -m pytest tests/test_kernels_attn_proj.py tests/test_unet_engine.py -m gpu -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>$o/bench.err | tail -1 > $o/bench.json
(cd /tmp && PIDM_NO_OVERLAP=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $o/prof.log 2>&1)
tail -3 $o/pytest.log; cut -c1-330 $o/bench.json; echo; grep -E "lap_|la_|layernorm" $o/prof/p_kernel_stats.csv | cut -c1-60,100-400 | head -20
