#!/bin/bash
# HBM traffic per kernel from the PMC counters, as MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE and WRITE_SIZE in
# SEPARATE --pmc passes (TCC slots), kernel-trace only; FETCH_SIZE is doubled on gfx950 (128-B requests tallied at 64 B).
# A 1 GiB device copy rides along in each pass as the calibration of both counters on a known byte count.
# usage (on the GPU box): tools/pmc_traffic.sh [per-GPU batch] [workload: darcy | mechanics | sampling]
# result: gpurun_out/pmc_traffic[_<workload>]/traffic.json -> profiles/pmc_traffic[_<workload>]_b<batch>.json (read by bench.py: roofline.traffic)
R=${GRAFT_REPO_ROOT:-/root/repo}; B=${1:-64}; W=${2:-darcy}
export TMPDIR=/tmp PYTHONPATH=$R; out=$R/gpurun_out/pmc_traffic; [ $W != darcy ] && out=${out}_$W; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 180 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -o p -- \
      python $R/bench.py --workload $W --steps 1 --warmup 1 --batch $B --no-cpu-baseline --no-roofline --no-alt --calib-copy > $out/$c.log 2>&1)
done
python $R/tools/pmc_traffic_report.py $out $B $W | tee $out/summary.txt
