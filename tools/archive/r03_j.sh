R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r03j}; rm -rf $o; mkdir -p $o
timeout 600 python -m pytest tests/test_kernels_attn_proj.py tests/test_unet_engine.py tests/test_training_step.py -m gpu -x -q 2>&1 | tail -2
for pp in 1 0; do
PIDM_LAP_SPLIT_PROJ=$pp timeout 600 python bench.py --no-cpu-baseline --no-alt --steps 40 2>$o/bench_pp$pp.err | tail -1 > $o/bench_pp$pp.json
python - $o/bench_pp$pp.json $pp <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("split_proj", sys.argv[2], d["value"], d["ms_per_step"])
PY
done
(cd /tmp && PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-roofline > $o/prof.log 2>&1)
find $o -name '*.db' -delete; find $o -name '*agent_info.csv' -delete; find $o -name '*kernel_trace.csv' -delete
python - $o/prof <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(f[0]))); N=25
print(f"kernel time {sum(float(r['TotalDurationNs']) for r in rows)/1e6/N:.3f} ms/step")
for r in rows:
    if 'lap_' in r['Name'] or 'copy_add' in r['Name'] or 'conv3x3_split' in r['Name']:
        print(f"{r['Name'].replace('void pidm::','').replace('pidm::','')[:50]:50s} calls/step={int(r['Calls'])/N:6.1f} avg_us={float(r['AverageNs'])/1e3:8.1f} ms/step={float(r['TotalDurationNs'])/1e6/N:7.3f}")
PY
