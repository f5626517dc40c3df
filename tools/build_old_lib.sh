# ab_old/libpidm_hip.so = the library with ONE source file taken from git HEAD (same-box A/B of a change: tools/ab_lib.sh)
# usage: tools/build_old_lib.sh k_norm.hip      (run here, after `make`: the other objects come from csrc/build)
set -e
f=$1; R=$(cd $(dirname $0)/.. && pwd); C=$R/physicsinformeddiffusionmodels_amd/csrc
mkdir -p $R/ab_old && git -C $R show HEAD:physicsinformeddiffusionmodels_amd/csrc/$f > $R/ab_old/$f
cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-inline-asm -ffp-contract=fast -I. -I$R/include -x hip -c $R/ab_old/$f -o $R/ab_old/$f.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab_old/libpidm_hip.so $(ls build/*.o | grep -v "/$f.o") $R/ab_old/$f.o -ldl
ls -la $R/ab_old/libpidm_hip.so
