# rocprofv3 kernel statistics of the Darcy step at batch 16 (where the step's fixed cost lives)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r05small}; rm -rf $o; mkdir -p $o
for b in 16; do
(cd /tmp && PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_b$b -o p -- python $R/bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-roofline > $o/prof_b$b.log 2>&1)
done
find $o -name '*.db' -delete; find $o -name '*agent_info.csv' -delete; find $o -name '*kernel_trace.csv' -delete
ls $o/prof_b16
