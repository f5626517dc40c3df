# GroupNorm apply kernels with the first pixels requested before the statistics are finalised: GPU parity + kernel times
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/gn; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_unet_engine.py tests/test_unet_config_sweep.py tests/test_training_step.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -2
for r in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-alt --no-roofline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"; done
timeout 300 python bench.py --batch 256 --no-cpu-baseline --no-alt --no-roofline --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench b256', d['value'], d['ms_per_step'])"
(cd /tmp && PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/p -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $o/log.txt 2>&1)
python - $o/p/p_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
steps=[int(r['Calls']) for r in rows if 'reduce_multi' in r['Name']][0]
print('steps',steps,'total ms',sum(float(r['TotalDurationNs']) for r in rows)/steps/1e6)
for r in rows:
    if any(k in r['Name'] for k in ('gn_','conv3x3_split_kernel<8, 0>','wgrad_split','lap_bwd','layernorm')):
        print(f"   {r['Name'][:56]:56s} {int(r['Calls'])/steps:5.1f} {float(r['TotalDurationNs'])/1e6/steps:7.3f}ms avg {float(r['AverageNs'])/1e3:7.1f} min {float(r['MinNs'])/1e3:6.1f} max {float(r['MaxNs'])/1e3:6.1f}")
PY
find $o -name '*.db' -delete; find $o -name '*_trace.csv' -delete
