#!/bin/bash
# round 6: HBM bytes of the Darcy loss kernels at batch 4096 (FETCH_SIZE / WRITE_SIZE, separate passes; 2 x FETCH_SIZE on gfx950)
R=${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
o=$R/gpurun_out/r06_x2; mkdir -p $o
for v in band full stream; do
  for c in FETCH_SIZE WRITE_SIZE; do
    unset PIDM_DARCY_FULL PIDM_DARCY_STREAM
    if [ $v = band ]; then export PIDM_DARCY_FULL=0; fi
    if [ $v = full ]; then export PIDM_DARCY_STREAM=0; fi
    (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $o/${v}_$c -o p -- python $R/tools/bench_darcy.py 4096 > $o/${v}_$c.log 2>&1)
  done
done
python - $o <<'PY'
import csv, glob, sys, collections
for v in ("band", "full", "stream"):
    m = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = collections.defaultdict(list)
        for f in glob.glob(f"{sys.argv[1]}/{v}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c and "darcy" in r["Kernel_Name"] and "finalize" not in r["Kernel_Name"]:
                    vals[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, x in vals.items(): m[c] = (k, sum(x) / len(x), len(x))
    print(v, m, "read MB (2x):", 2 * m["FETCH_SIZE"][1] * 1024 / 1e6, "write MB:", m["WRITE_SIZE"][1] * 1024 / 1e6)
PY
rm -rf $o/*_SIZE
