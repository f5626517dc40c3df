# round 4, call O: rocprofv3 kernel statistics of the Darcy step at batch 64 and 256 (overlap off)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/${1:-r04o}; mkdir -p $O
for b in 64 256; do
  st=20; [ $b = 256 ] && st=8
  (cd /tmp && PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$b -o p -- python $R/bench.py --batch $b --steps $st --warmup 5 --no-cpu-baseline --no-alt --no-roofline > $O/prof_$b.log 2>&1)
  cp $(find $O/prof_$b -name 'p_kernel_stats.csv' | head -1) $O/kernel_stats_b$b.csv
done
find $O -name '*.db' -delete; find $O -name '*agent_info.csv' -delete; find $O -name '*kernel_trace.csv' -delete
ls $O
