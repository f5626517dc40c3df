# kernel statistics of the three bench workloads (overlap off so kernels have their own durations)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r02n}; rm -rf $o; mkdir -p $o
for w in darcy mechanics sampling; do
  st=20; [ $w = mechanics ] && st=6
  (cd /tmp && PIDM_NO_OVERLAP=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$w -o p -- python $R/bench.py --workload $w --steps $st --warmup 3 --no-cpu-baseline --no-roofline --no-alt > $o/prof_$w.log 2>&1)
  tail -1 $o/prof_$w.log | cut -c1-200
done
