# round 4, call Q: half-wave exchanges by v_permlane32_swap instead of ds_bpermute in the attention kernels and conv epilogues - A/B against the previous library
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/${1:-r04q}; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_attn.py tests/test_kernels_attn_proj.py tests/test_unet_engine.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
for rep in 1 2; do
for v in old new; do
  lib=""; [ $v = old ] && lib="PIDM_LIBRARY=$R/tools/_abl/libpidm_old.so"
  for b in 64 256; do
    echo "#### $v batch $b"
    env $lib timeout 600 python bench.py --batch $b --steps 30 --warmup 10 --no-cpu-baseline --no-alt --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
done > $O/step.txt 2>&1
cat $O/step.txt
