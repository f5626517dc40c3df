R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/lapp; rm -rf $o; mkdir -p $o
for g in auto 1; do for cfg in "64 64 4 32" "64 32 4 64"; do
[ $g = 1 ] && export PIDM_LAP_GROUPS=1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/p$g -o p -- python $R/tools/bench_lap.py $cfg > $o/log.txt 2>&1)
grep "^lap" $o/log.txt
f=$(find $o/p$g -name '*kernel_stats.csv' | head -1); python - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'lap_' in r['Name'] or 'reduce' in r['Name']: print(f"   {r['Name'][:44]:44s} {int(r['Calls']):4d} {float(r['AverageNs'])/1e3:8.1f} us")
PY
rm -rf $o/p$g
done; done
