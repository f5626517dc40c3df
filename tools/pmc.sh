#!/bin/bash
# usage: tools/pmc.sh <outdir-name> <python script + args...>   (three separate --pmc passes, kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}; name=$1; shift
export TMPDIR=/tmp; mkdir -p $R/gpurun_out/pmc_$name
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM"; do
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_$name/g$i -o p -- python "$@" > $R/gpurun_out/pmc_$name/g$i.log 2>&1)
done
ls $R/gpurun_out/pmc_$name/*
