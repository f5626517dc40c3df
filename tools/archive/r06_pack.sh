#!/bin/bash
# round 6: pack_multi_kernel with 16-byte moves for whole tiles - the build before against HEAD, one box, alternating legs
R=${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_kernels_conv.py tests/test_unet_engine.py -x -q -m gpu 2>&1 | tail -1
for rep in 1 2; do
  for l in before head; do
    if [ $l = before ]; then export PIDM_LIBRARY=$R/tools/ab/libpidm_pack_before.so; else unset PIDM_LIBRARY; fi
    python bench.py --workload mechanics --steps 10 --warmup 4 --no-cpu-baseline --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$l mechanics:', d['ms_per_step'], 'ms')"
    python bench.py --batch 64 --steps 30 --warmup 8 --no-cpu-baseline --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$l darcy 64:', d['ms_per_step'], 'ms')"
  done
done
unset PIDM_LIBRARY
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_pack -o p -- python $R/bench.py --workload mechanics --steps 6 --warmup 3 --no-cpu-baseline --no-alt --no-roofline > /dev/null 2>&1
grep -E "pack_multi|Name" $R/gpurun_out/r06_pack/p_kernel_stats.csv | cut -c1-160
rm -rf $R/gpurun_out/r06_pack
