# GPU pass: projected-attention parity tests, engine goldens in all attention forms, bench A/B (qkv form vs projected), kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r02d}; rm -rf $o; mkdir -p $o
timeout 600 python -m pytest tests/test_kernels_attn_proj.py tests/test_unet_engine.py tests/test_training_step.py tests/test_gpu_fullsize.py -m gpu -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
PIDM_NO_LAP=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>$o/bench.err | tail -1 > $o/bench_qkv.json
PIDM_PROF_DUMP=$o/shape_table.txt timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>>$o/bench.err | tail -1 > $o/bench_proj.json
timeout 300 python bench.py --steps 20 --warmup 5 --batch 256 --no-cpu-baseline --no-roofline 2>>$o/bench.err | tail -1 > $o/bench_proj_b256.json
(cd /tmp && PIDM_NO_OVERLAP=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $o/prof.log 2>&1)
tail -4 $o/pytest.log; cut -c1-330 $o/bench_qkv.json; echo; cut -c1-330 $o/bench_proj.json; echo; cut -c1-200 $o/bench_proj_b256.json; echo; head -25 $o/prof/p_kernel_stats.csv | cut -c1-150
