#!/bin/bash
# rocprofv3 kernel stats of the default bench command; prints per-kernel ms/step for kernels matching $1 (regex)
R=${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp PYTHONPATH=$R; out=$R/gpurun_out/pb; rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $out/log.txt 2>&1)
grep -o '"ms_per_step": [0-9.]*' $out/log.txt
python - "$out/p_kernel_stats.csv" "${1:-.}" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1]))); N=25
tot=sum(float(r['TotalDurationNs']) for r in rows)/1e6/N
print(f"kernel time {tot:.3f} ms/step")
for r in rows:
    n=r['Name'].replace('void pidm::','').replace('pidm::','')
    if re.search(sys.argv[2], n):
        print(f"{n[:70]:70s} calls/step={int(r['Calls'])/N:6.1f} avg_us={float(r['AverageNs'])/1e3:8.1f} ms/step={float(r['TotalDurationNs'])/1e6/N:6.3f}")
PY
