# the four PMC traffic passes only (profiles/pmc_traffic_*.json)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; o=$R/gpurun_out/${1:-r03pmc}; rm -rf $o; mkdir -p $o
rm -rf gpurun_out/pmc_traffic gpurun_out/pmc_traffic_mechanics gpurun_out/pmc_traffic_sampling
bash tools/pmc_traffic.sh 64 darcy > $o/b64.log 2>&1; cp gpurun_out/pmc_traffic/traffic.json $o/pmc_traffic_b64.json; rm -rf gpurun_out/pmc_traffic
bash tools/pmc_traffic.sh 256 darcy > $o/b256.log 2>&1; cp gpurun_out/pmc_traffic/traffic.json $o/pmc_traffic_b256.json; rm -rf gpurun_out/pmc_traffic
bash tools/pmc_traffic.sh 32 mechanics > $o/mech.log 2>&1; cp gpurun_out/pmc_traffic_mechanics/traffic.json $o/pmc_traffic_mechanics_b32.json; rm -rf gpurun_out/pmc_traffic_mechanics
bash tools/pmc_traffic.sh 1024 sampling > $o/samp.log 2>&1; cp gpurun_out/pmc_traffic_sampling/traffic.json $o/pmc_traffic_sampling_b1024.json; rm -rf gpurun_out/pmc_traffic_sampling
ls -la $o
