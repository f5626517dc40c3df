R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/lapp; rm -rf $o; mkdir -p $o
python -m pytest tests/test_kernels_attn_proj.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
for cfg in "64 64 8 32" "64 32 8 64" "256 64 8 32"; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/p -o p -- python $R/tools/bench_lap.py $cfg > $o/log.txt 2>&1)
grep "^lap" $o/log.txt
f=$(find $o/p -name '*kernel_stats.csv' | head -1); python - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'lap_' in r['Name']: print(f"   {r['Name'][:44]:44s} {int(r['Calls']):4d} {float(r['AverageNs'])/1e3:8.1f} us")
PY
rm -rf $o/p
done
