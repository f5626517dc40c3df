#!/bin/bash
# round 6: one-workgroup-per-sample Darcy kernels against the band kernel (tools/bench_darcy.py), and their parity tests on the GPU
mkdir -p gpurun_out/r06_t
python -m pytest tests/test_kernels_darcy.py -x -q -m gpu > gpurun_out/r06_t/tests.log 2>&1
tail -3 gpurun_out/r06_t/tests.log
for rep in 1 2; do
  PIDM_DARCY_FULL=0 python tools/bench_darcy.py 512 1024 4096 2>&1 | grep "B=" | sed "s/^/band     /"
  PIDM_DARCY_STREAM=0 PIDM_DARCY_RES_T=0 python tools/bench_darcy.py 512 1024 4096 2>&1 | grep "B=" | sed "s/^/full direct  /"
  PIDM_DARCY_STREAM=0 python tools/bench_darcy.py 512 1024 4096 2>&1 | grep "B=" | sed "s/^/full res_t  /"
  python tools/bench_darcy.py 512 1024 4096 2>&1 | grep "B=" | sed "s/^/stream   /"
done
