# PMC traffic of the four workloads (tools/pmc_traffic.sh) in one gpurun call -> profiles/pmc_traffic_*.json
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r05pmc}; rm -rf $o; mkdir -p $o/pmc
rm -rf gpurun_out/pmc_traffic gpurun_out/pmc_traffic_mechanics gpurun_out/pmc_traffic_sampling
bash tools/pmc_traffic.sh 64 darcy > $o/pmc_b64.log 2>&1; cp gpurun_out/pmc_traffic/traffic.json $o/pmc/pmc_traffic_b64.json; rm -rf gpurun_out/pmc_traffic
bash tools/pmc_traffic.sh 256 darcy > $o/pmc_b256.log 2>&1; cp gpurun_out/pmc_traffic/traffic.json $o/pmc/pmc_traffic_b256.json; rm -rf gpurun_out/pmc_traffic
bash tools/pmc_traffic.sh 32 mechanics > $o/pmc_mech.log 2>&1; cp gpurun_out/pmc_traffic_mechanics/traffic.json $o/pmc/pmc_traffic_mechanics_b32.json; rm -rf gpurun_out/pmc_traffic_mechanics
bash tools/pmc_traffic.sh 1024 sampling > $o/pmc_samp.log 2>&1; cp gpurun_out/pmc_traffic_sampling/traffic.json $o/pmc/pmc_traffic_sampling_b1024.json; rm -rf gpurun_out/pmc_traffic_sampling
python - $o <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/pmc/*.json")):
    d=json.load(open(f)); print(f.split('/')[-1], {k:(v['launches'], round(v['hbm_bytes_per_step']/1e9,3)) for k,v in d.items() if isinstance(v,dict) and 'hbm_bytes_per_step' in v}, d['calibration_1GiB_copy'])
PY
