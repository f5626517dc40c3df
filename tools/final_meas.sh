# end-of-round measurement batch (one gpurun call): PMC traffic -> bench lines (batch 64, 256) -> rocprofv3 kernel stats -> secondary benches
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/fin; rm -rf $o; mkdir -p $o
bash tools/pmc_traffic.sh 64 > $o/pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic/summary.txt $o/ 2>/dev/null; [ -s gpurun_out/pmc_traffic/traffic.json ] && cp gpurun_out/pmc_traffic/traffic.json profiles/pmc_traffic_b64.json; cp profiles/pmc_traffic_b64.json $o/pmc_traffic_b64.json
python bench.py --steps 20 --warmup 5 2>$o/bench.err | tail -1 > $o/bench.json
python bench.py --steps 20 --warmup 5 --batch 256 --no-cpu-baseline 2>>$o/bench.err | tail -1 > $o/bench_b256.json
(cd /tmp && PIDM_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/prof.log 2>&1)
python tools/bench_secondary.py > $o/secondary.json 2>$o/secondary.err
python tools/bench_attn.py > $o/attn.txt 2>&1
cut -c1-300 $o/bench.json; cut -c1-200 $o/bench_b256.json; tail -3 $o/secondary.json | cut -c1-300; ls $o $o/prof | head -30
