# PMC traffic passes for the workloads / batch sizes that had none (mechanics b32, sampling b1024, darcy b256)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/pmc_traffic.sh 32 mechanics > gpurun_out/pmc_mech.log 2>&1
bash tools/pmc_traffic.sh 1024 sampling > gpurun_out/pmc_samp.log 2>&1
mkdir -p gpurun_out/pmc_keep; cp gpurun_out/pmc_traffic_mechanics/traffic.json gpurun_out/pmc_keep/pmc_traffic_mechanics_b32.json; cp gpurun_out/pmc_traffic_sampling/traffic.json gpurun_out/pmc_keep/pmc_traffic_sampling_b1024.json
rm -rf gpurun_out/pmc_traffic_mechanics/*SIZE gpurun_out/pmc_traffic_sampling/*SIZE
rm -rf gpurun_out/pmc_traffic; bash tools/pmc_traffic.sh 256 darcy > gpurun_out/pmc_b256.log 2>&1
cp gpurun_out/pmc_traffic/traffic.json gpurun_out/pmc_keep/pmc_traffic_b256.json; rm -rf gpurun_out/pmc_traffic/*SIZE
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/pmc_keep/*.json')):
    d=json.load(open(f)); print(f, {k:(round(v['hbm_bytes_per_step']/1e9,3), v['launches']) for k,v in d.items() if isinstance(v,dict) and 'hbm_bytes_per_step' in v}, d['calibration_1GiB_copy'])
PY
tail -3 gpurun_out/pmc_mech.log gpurun_out/pmc_samp.log gpurun_out/pmc_b256.log | cut -c1-300
