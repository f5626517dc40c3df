# round-5 measurement batch (one gpurun call): GPU tests -> PMC of the split kernels -> PMC traffic (4 workloads) -> bench lines ->
# gradient-exchange legs (darcy, mechanics; torch.distributed RCCL and the C-ABI communicator) -> rocprofv3 kernel stats -> batch sweep
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r05fin}; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log; grep -E "passed|failed|rc=" $o/pytest.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; tail -1 $o/smoke.log
# ---- two ranks on this one GPU over gloo (debug mode of bench.py): the N > 1 control flow on hardware, and the exchange's negotiation for
# real - RCCL refuses two ranks on one device, so every rank must come back to torch.distributed together (exchange.collective_note)
PIDM_BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 2 --steps 10 --warmup 4 --no-cpu-baseline --no-alt --no-roofline 2>$o/share_gpu.err | tail -1 > $o/bench_two_ranks_one_gpu.json
python -c "
import json; d=json.load(open('$o/bench_two_ranks_one_gpu.json')); print('2 ranks / 1 GPU:', d['value'], d['n_gpus'], d['exchange'], d['per_rank'])" 2>&1 | cut -c1-900
# ---- PMC counters of the split-form 3x3 kernels (three separate --pmc passes each, kernel-trace only) ----
bash tools/pmc.sh sp64 $R/tools/bench_one.py 64 32 0 32 3 1 1 0 64 3 > /dev/null 2>&1
bash tools/pmc.sh sp16 $R/tools/bench_one.py 16 128 0 128 3 1 1 0 64 3 > /dev/null 2>&1
bash tools/pmc.sh sp8 $R/tools/bench_one.py 8 256 0 256 3 1 1 0 64 3 > /dev/null 2>&1
bash tools/pmc.sh sp64b256 $R/tools/bench_one.py 64 32 0 32 3 1 1 0 256 3 > /dev/null 2>&1
for n in sp64 sp16 sp8 sp64b256; do echo "#### $n"; python tools/pmc_report.py gpurun_out/pmc_$n conv3x3_rs conv3x3_split conv_wgrad_rs conv_wgrad_split; done > $o/pmc_split_kernels.txt
rm -rf gpurun_out/pmc_sp64 gpurun_out/pmc_sp16 gpurun_out/pmc_sp8 gpurun_out/pmc_sp64b256
grep -E "####|==|waves=" $o/pmc_split_kernels.txt | cut -c1-260
# ---- PMC traffic, all four workloads ----
mkdir -p $o/pmc
rm -rf gpurun_out/pmc_traffic gpurun_out/pmc_traffic_mechanics gpurun_out/pmc_traffic_sampling
bash tools/pmc_traffic.sh 64 darcy > $o/pmc_b64.log 2>&1; cp gpurun_out/pmc_traffic/traffic.json $o/pmc/pmc_traffic_b64.json; cp gpurun_out/pmc_traffic/summary.txt $o/pmc/summary_b64.txt; rm -rf gpurun_out/pmc_traffic
bash tools/pmc_traffic.sh 256 darcy > $o/pmc_b256.log 2>&1; cp gpurun_out/pmc_traffic/traffic.json $o/pmc/pmc_traffic_b256.json; cp gpurun_out/pmc_traffic/summary.txt $o/pmc/summary_b256.txt; rm -rf gpurun_out/pmc_traffic
bash tools/pmc_traffic.sh 32 mechanics > $o/pmc_mech.log 2>&1; cp gpurun_out/pmc_traffic_mechanics/traffic.json $o/pmc/pmc_traffic_mechanics_b32.json; rm -rf gpurun_out/pmc_traffic_mechanics
bash tools/pmc_traffic.sh 1024 sampling > $o/pmc_samp.log 2>&1; cp gpurun_out/pmc_traffic_sampling/traffic.json $o/pmc/pmc_traffic_sampling_b1024.json; rm -rf gpurun_out/pmc_traffic_sampling
cp $o/pmc/*.json profiles/ 2>/dev/null
# ---- bench lines ----
timeout 900 python bench.py --steps 20 --warmup 5 2>$o/bench.err | tail -1 > $o/bench.json
timeout 600 python bench.py --workload mechanics --steps 10 --warmup 4 2>>$o/bench.err | tail -1 > $o/bench_mechanics.json
timeout 600 python bench.py --workload sampling --steps 20 --warmup 5 2>>$o/bench.err | tail -1 > $o/bench_sampling.json
# ---- gradient exchange on one GPU (world size 1: the RCCL path, the side stream and the phase events run for real) ----
# (default collective on GPUs since round 5: the C-ABI communicator, negotiated and self-checked; PIDM_DP_NATIVE=0: torch.distributed)
PIDM_BENCH_FORCE_EXCHANGE=1 timeout 600 python bench.py --no-cpu-baseline --no-alt --no-roofline --steps 20 2>>$o/bench.err | tail -1 > $o/force_exchange_darcy.json
PIDM_BENCH_FORCE_EXCHANGE=1 PIDM_DP_NATIVE=0 timeout 600 python bench.py --no-cpu-baseline --no-alt --no-roofline --steps 20 2>>$o/bench.err | tail -1 > $o/force_exchange_darcy_torch.json
PIDM_BENCH_FORCE_EXCHANGE=1 timeout 600 python bench.py --workload mechanics --no-cpu-baseline --no-alt --no-roofline --steps 10 --warmup 4 2>>$o/bench.err | tail -1 > $o/force_exchange_mechanics.json
PIDM_BENCH_FORCE_EXCHANGE=1 PIDM_DP_NATIVE=0 timeout 600 python bench.py --workload mechanics --no-cpu-baseline --no-alt --no-roofline --steps 10 --warmup 4 2>>$o/bench.err | tail -1 > $o/force_exchange_mechanics_torch.json
# ---- rocprofv3 kernel statistics (overlap off: a kernel that shares the chip has no duration of its own) ----
for w in darcy mechanics sampling; do
  st=20; [ $w = mechanics ] && st=6
  (cd /tmp && PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$w -o p -- python $R/bench.py --workload $w --steps $st --warmup 5 --no-cpu-baseline --no-alt > $o/prof_$w.log 2>&1)
done
(cd /tmp && PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_darcy_b256 -o p -- python $R/bench.py --batch 256 --steps 8 --warmup 3 --no-cpu-baseline --no-alt > $o/prof_darcy_b256.log 2>&1)
find $o -name '*.db' -delete; find $o -name '*agent_info.csv' -delete; find $o -name '*kernel_trace.csv' -delete
# ---- the step against the per-GPU batch ----
for b in 16 32 64 128 256 512; do
st=30; [ $b -ge 256 ] && st=10
timeout 600 python bench.py --batch $b --no-cpu-baseline --no-alt --no-roofline --steps $st --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
b=$b
print(f'batch {b:4d}: {d[\"value\"]:8.1f} samples/s  {d[\"ms_per_step\"]:8.3f} ms/step  step_flop_fraction {b*11.916e9/(d[\"ms_per_step\"]*1e-3)/157.3e12:.3f}  kernels per step {d[\"launches\"][\"kernels_inside_graphs_per_step\"] + d[\"launches\"][\"kernels_enqueued_one_by_one_per_step\"]:.0f}')"
done | tee $o/batch_sweep.txt
python - $o <<'PY'
import json,sys
o=sys.argv[1]
for f in ("bench.json","bench_mechanics.json","bench_sampling.json","force_exchange_darcy.json","force_exchange_darcy_torch.json","force_exchange_mechanics.json","force_exchange_mechanics_torch.json"):
    try:
        d=json.load(open(f"{o}/{f}"))
    except Exception as e:
        print(f, "unreadable", e); continue
    r=d.get("roofline") or {}
    print(f, d["value"], d["unit"], d["ms_per_step"], "sff", d.get("step_flop_fraction"), {k:(d.get(k) or {}).get("value") for k in ("fp32_mfma_only","eager_scalars","dropin_main_py","north_star_b256","mechanics_b32","sampling_b1024")}, {k:r.get(k) for k in ("achieved","frac","frac_bf16_pipe","traffic","step_traffic_over_contract")}, "exchange", d.get("exchange"))
    if d.get("north_star_b256"): print("   b256:", {k:v for k,v in d["north_star_b256"].items() if k!="what"})
PY
ls $o
