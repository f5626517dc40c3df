# round 4, call B: variants of conv_wgrad_rs_kernel (PIDM_WGRAD_RS_VAR = 0 / 1 / 2) and the same file built with -fno-slp-vectorize
# (v_pk_add_f32 beside MFMAs is an anti-lever per MI355X_MICROARCH.md), per shape at batch 64 / 256, one box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/${1:-r04b}; mkdir -p $O
C=$R/physicsinformeddiffusionmodels_amd/csrc
# alternate build of the one file
( cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize -x hip -c k_wgrad_rs.hip -o /tmp/k_wgrad_rs_noslp.o \
  && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libpidm_noslp.so $(ls build/*.o | grep -v k_wgrad_rs) /tmp/k_wgrad_rs_noslp.o ) 2>&1 | tail -3
ls -la /tmp/libpidm_noslp.so
SH="64,32,0,32,3,1,1,0;64,32,32,32,3,1,1,0;32,64,0,64,3,1,1,0;16,128,0,128,3,1,1,0;8,256,0,256,3,1,1,0"
for b in 64 256; do
  for lib in "" /tmp/libpidm_noslp.so; do
    for v in 0 1 2; do
      echo "== batch $b lib=${lib:-default} PIDM_WGRAD_RS_VAR=$v"
      PIDM_LIBRARY=$lib BENCH_CONV_SHAPES="$SH" PIDM_WGRAD_RS_VAR=$v timeout 300 python tools/bench_conv.py $b 2>&1 | grep -E "^H=|TOTAL" | sed 's/| fwd.*| wgrad/| wgrad/'
    done
  done
done > $O/wgrad_var.txt 2>&1
cat $O/wgrad_var.txt
PIDM_WGRAD_RS_VAR=2 timeout 600 python -m pytest tests/test_kernels_conv.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
PIDM_WGRAD_RS_VAR=1 timeout 600 python -m pytest tests/test_kernels_conv.py -m gpu -x -q -k "split_forms or accurate" 2>&1 | grep -E "passed|failed"
