#!/bin/bash
# usage: tools/pmc_mem.sh <outdir-name> <absolute python script + args...>   vector-memory path counters (TA / TCP / TD), own passes
# NOTE (round 1, ROCm 7.2 on this pool): the TA_* and TCP_* groups abort rocprofv3 (signal 6) and then sit in their 200 s timeout;
# only the third group (TD_TD_BUSY, TCP_GATE_EN*, TCP_TA_TCP_STATE_READ) returned data.  Budget ~7 GPU-minutes if you run it as is.
R=${GRAFT_REPO_ROOT:-/root/repo}; name=$1; shift
export TMPDIR=/tmp; mkdir -p $R/gpurun_out/pmc_$name
i=0
for grp in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_$name/m$i -o p -- python "$@" > $R/gpurun_out/pmc_$name/m$i.log 2>&1)
done
python - $R/gpurun_out/pmc_$name <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{sys.argv[1]}/m*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void pidm::", "").replace("pidm::", "")[:40]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if not k.startswith(("la_", "conv_")): continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print("==", k, "  ".join(f"{c.replace('_sum','')}={val:.3g}" for c, val in sorted(m.items())))
PY
