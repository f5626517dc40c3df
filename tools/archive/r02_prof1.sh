# kernel statistics of one bench workload
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r02q}; w=${2:-darcy}; rm -rf $o; mkdir -p $o
(cd /tmp && PIDM_NO_OVERLAP=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$w -o p -- python $R/bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-alt > $o/prof_$w.log 2>&1)
tail -1 $o/prof_$w.log | cut -c1-200
find $o -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $o/kernel_stats_$w.csv
find $o -name "*.csv" ! -name "kernel_stats_*" -delete; find $o -name "*.db" -delete
