R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r02t}; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_kernels_attn_proj.py tests/test_unet_engine.py tests/test_training_step.py tests/test_gpu_fullsize.py -m gpu -q -x > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
tail -3 $o/pytest.log
bash tools/r02_prof1.sh ${1:-r02t}_p darcy
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-200
