# round-3 measurement batch (one gpurun call): GPU tests -> PMC traffic -> bench lines (3 workloads) -> rocprofv3 kernel stats -> gaps
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r03fin}; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log; tail -3 $o/pytest.log
bash tools/pmc_traffic.sh 64 > $o/pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic/summary.txt $o/ 2>/dev/null; [ -s gpurun_out/pmc_traffic/traffic.json ] && cp gpurun_out/pmc_traffic/traffic.json profiles/pmc_traffic_b64.json; cp profiles/pmc_traffic_b64.json $o/pmc_traffic_b64.json
rm -rf gpurun_out/pmc_traffic/FETCH_SIZE gpurun_out/pmc_traffic/WRITE_SIZE
# the other workloads / batch sizes (profiles/pmc_traffic_b256.json, _mechanics_b32, _sampling_b1024)
mkdir -p $o/pmc
cp gpurun_out/pmc_traffic/traffic.json $o/pmc/pmc_traffic_b64.json
rm -rf gpurun_out/pmc_traffic; bash tools/pmc_traffic.sh 256 darcy > $o/pmc_b256.log 2>&1; cp gpurun_out/pmc_traffic/traffic.json $o/pmc/pmc_traffic_b256.json; rm -rf gpurun_out/pmc_traffic
bash tools/pmc_traffic.sh 32 mechanics > $o/pmc_mech.log 2>&1; cp gpurun_out/pmc_traffic_mechanics/traffic.json $o/pmc/pmc_traffic_mechanics_b32.json; rm -rf gpurun_out/pmc_traffic_mechanics
bash tools/pmc_traffic.sh 1024 sampling > $o/pmc_samp.log 2>&1; cp gpurun_out/pmc_traffic_sampling/traffic.json $o/pmc/pmc_traffic_sampling_b1024.json; rm -rf gpurun_out/pmc_traffic_sampling
cp $o/pmc/*.json profiles/
timeout 600 python bench.py 2>$o/bench.err | tail -1 > $o/bench.json
timeout 600 python bench.py --workload mechanics --steps 10 --warmup 4 2>>$o/bench.err | tail -1 > $o/bench_mechanics.json
timeout 600 python bench.py --workload sampling --steps 20 --warmup 5 2>>$o/bench.err | tail -1 > $o/bench_sampling.json
for w in darcy mechanics sampling; do
  st=20; [ $w = mechanics ] && st=6
  (cd /tmp && PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$w -o p -- python $R/bench.py --workload $w --steps $st --warmup 5 --no-cpu-baseline --no-alt > $o/prof_$w.log 2>&1)
done
(cd /tmp && PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_darcy_b256 -o p -- python $R/bench.py --batch 256 --steps 8 --warmup 3 --no-cpu-baseline --no-alt > $o/prof_darcy_b256.log 2>&1)
find $o -name '*.db' -delete; find $o -name '*agent_info.csv' -delete; find $o -name '*kernel_trace.csv' -delete
python - $o <<'PY'
import json,sys
o=sys.argv[1]
for f in ("bench.json","bench_mechanics.json","bench_sampling.json"):
    try:
        d=json.load(open(f"{o}/{f}"))
    except Exception as e:
        print(f, "unreadable", e); continue
    r=d.get("roofline") or {}
    print(f, d["value"], d["unit"], d["ms_per_step"], {k:(d.get(k) or {}).get("value") for k in ("fp32_mfma_only","eager_scalars","dropin_main_py","north_star_b256")}, {k:r.get(k) for k in ("achieved","frac","frac_bf16_pipe","traffic","step_flop_fraction","step_hbm_fraction")})
    if d.get("north_star_b256"): print("   b256:", d["north_star_b256"])
    if d.get("residual_only"): print("   residual_only:", {k:v for k,v in d["residual_only"].items() if k!="what"})
PY
grep -A3 '"conv"' $o/pmc_traffic_b64.json | head -8
ls $o
