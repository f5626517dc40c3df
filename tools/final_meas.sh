# end-of-round measurement batch (one gpurun call): PMC traffic -> bench lines (3 workloads, batch 256) -> rocprofv3 kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-fin}; rm -rf $o; mkdir -p $o
bash tools/pmc_traffic.sh 64 > $o/pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic/summary.txt $o/ 2>/dev/null; [ -s gpurun_out/pmc_traffic/traffic.json ] && cp gpurun_out/pmc_traffic/traffic.json profiles/pmc_traffic_b64.json; cp profiles/pmc_traffic_b64.json $o/pmc_traffic_b64.json
python bench.py 2>$o/bench.err | tail -1 > $o/bench.json
python bench.py --ema --no-cpu-baseline --no-roofline --no-alt 2>>$o/bench.err | tail -1 > $o/bench_ema.json
python bench.py --eager-scalars --no-cpu-baseline --no-roofline --no-alt 2>>$o/bench.err | tail -1 > $o/bench_eager_scalars.json
python bench.py --batch 256 --steps 20 --warmup 5 --no-cpu-baseline 2>>$o/bench.err | tail -1 > $o/bench_b256.json
python bench.py --workload mechanics --steps 10 --warmup 3 2>>$o/bench.err | tail -1 > $o/bench_mechanics.json
python bench.py --workload sampling --steps 20 --warmup 5 2>>$o/bench.err | tail -1 > $o/bench_sampling.json
for w in darcy mechanics sampling; do
  st=20; [ $w = mechanics ] && st=6
  (cd /tmp && PIDM_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$w -o p -- python $R/bench.py --workload $w --steps $st --warmup 5 --no-cpu-baseline --no-alt > $o/prof_$w.log 2>&1)
done
bash tools/archive/r02_gaps.sh ${1:-fin}_gaps 1 > $o/gaps.txt 2>&1
find $o -name '*.db' -delete; find $o -name '*agent_info.csv' -delete
cut -c1-300 $o/bench.json; cut -c1-200 $o/bench_b256.json; cut -c1-200 $o/bench_mechanics.json; cut -c1-200 $o/bench_sampling.json; ls $o
