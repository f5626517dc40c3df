#!/bin/bash
# per-launch durations of the normalisation kernels grouped by grid (= level): rocprofv3 --kernel-trace of a short bench.py run
# usage: tools/gn_by_shape.sh <out-name> [batch] [pattern,pattern,...] [workload]   (default patterns: gn_,layernorm; 'all' = every kernel;
# workload darcy (default) / mechanics / sampling; batch 0 = the workload's default)
R=${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; o=$R/gpurun_out/$1; mkdir -p $o; b=${2:-64}; export PIDM_SHAPE_PATTERNS=${3:-gn_,layernorm}
(cd /tmp && PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $o/trace -o p -- python $R/bench.py $( [ "$b" != 0 ] && echo --batch $b ) --workload ${4:-darcy} --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-roofline > $o/run.log 2>&1)
python - $o <<'PY'
import csv, glob, os, sys, collections
f = glob.glob(f"{sys.argv[1]}/trace/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    pats = os.environ.get("PIDM_SHAPE_PATTERNS", "gn_,layernorm").split(",")
    if pats != ["all"] and not any(k in n for k in pats): continue
    name = n.split("(")[0].replace("void pidm::", "").replace("pidm::", "").replace("void ", "")[:44]
    agg[(name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k, v in sorted(agg.items()):
    v.sort()
    print(f"{k[0]:44s} grid {k[1]:5d} x {k[2]:4d}  n {len(v):5d}  median {v[len(v)//2]:7.2f} us  min {v[0]:7.2f}  sum/step {sum(v)/13:8.1f} us")
    tot += sum(v) / 13
print("total per step", round(tot, 1), "us")
PY
rm -rf $o/trace
