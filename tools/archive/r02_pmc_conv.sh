R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
PIDM_CONV_STREAM=1 bash tools/pmc.sh s1_16 $R/tools/bench_one.py 16 128 0 128 3 1 1 0 64 3 > /dev/null 2>&1
for n in s1_16; do echo "#### $n"; python tools/pmc_report.py gpurun_out/pmc_$n conv_igemm conv3x3; done
