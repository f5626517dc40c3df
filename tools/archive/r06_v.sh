#!/bin/bash
# round 6: PMC counters of the Darcy loss kernel at batch 4096 - band kernel, one workgroup per sample (512 / 256 threads)
R=${GRAFT_REPO_ROOT:-/root/repo}
o=gpurun_out/r06_v; mkdir -p $o

PIDM_DARCY_FULL=0 bash tools/pmc.sh dband $R/tools/bench_darcy.py 4096 > /dev/null 2>&1
PIDM_DARCY_FULL_NT=512 bash tools/pmc.sh dfull512 $R/tools/bench_darcy.py 4096 > /dev/null 2>&1
PIDM_DARCY_FULL_NT=1024 bash tools/pmc.sh dfull1024 $R/tools/bench_darcy.py 4096 > /dev/null 2>&1
for n in dband dfull512 dfull1024; do echo "#### $n"; python tools/pmc_report.py gpurun_out/pmc_$n darcy; done > $o/pmc_darcy.txt
rm -rf gpurun_out/pmc_dband gpurun_out/pmc_dfull512 gpurun_out/pmc_dfull1024
cat $o/pmc_darcy.txt | cut -c1-400
