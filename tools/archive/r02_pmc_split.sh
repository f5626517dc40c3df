# PMC counters (three separate --pmc passes, kernel-trace only) of the split-form 3x3 kernels on the 64x64 32->32 and 16x16 128->128 layers
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
bash tools/pmc.sh sp64 $R/tools/bench_one.py 64 32 0 32 3 1 1 0 64 3 > /dev/null 2>&1
bash tools/pmc.sh sp16 $R/tools/bench_one.py 16 128 0 128 3 1 1 0 64 3 > /dev/null 2>&1
for n in sp64 sp16; do echo "#### $n"; python tools/pmc_report.py gpurun_out/pmc_$n conv3x3_split conv_wgrad_split; done
find gpurun_out/pmc_sp64 gpurun_out/pmc_sp16 -name "*.db" -delete
