# round 4, call D: GroupNorm-2 backward sums from the producing dgrad epilogue (A/B: PIDM_NO_BN2_EPILOGUE=1), knob snapshot, GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/${1:-r04d}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; grep -E "passed|failed|error" $O/gpu_suite.log | tail -3
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export PIDM_NO_BN2_EPILOGUE=1; else unset PIDM_NO_BN2_EPILOGUE; fi
  timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('darcy no_bn2=$v', d['value'], d['ms_per_step'], 'b256', d['north_star_b256']['value'], d['north_star_b256']['ms_per_step'], d['north_star_b256']['step_flop_fraction'], 'launches', d['launches']['kernels_inside_graphs_per_step'], 'mech', d.get('mechanics_b32',{}).get('value'), 'samp', d.get('sampling_b1024',{}).get('value'))"
done 2>&1 | tee $O/step_ab.txt
unset PIDM_NO_BN2_EPILOGUE
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
