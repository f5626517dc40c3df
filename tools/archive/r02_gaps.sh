# kernel timeline of the darcy bench (overlap off): idle gaps between consecutive kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r02g}; rm -rf $o; mkdir -p $o
(cd /tmp && PIDM_NO_OVERLAP=${2:-1} timeout 400 rocprofv3 --kernel-trace --output-format csv -d $o/prof -o p -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-alt > $o/prof.log 2>&1)
f=$(find $o -name "*kernel_trace.csv" | head -1); python - $f <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 8 steps: find step boundaries by the optimizer kernel
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "clip_adam" in n]
lo, hi = idx[-9] + 1, idx[-1] + 1
seg = rows[lo:hi]
nsteps = 8
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
gaps = collections.defaultdict(lambda: [0, 0])
tot_gap = 0
prev_end = int(seg[0]["End_Timestamp"])
for a, b in zip(seg[:-1], seg[1:]):
    g = int(b["Start_Timestamp"]) - max(prev_end, int(a["End_Timestamp"]))
    prev_end = max(prev_end, int(a["End_Timestamp"]))
    if g > 0:
        tot_gap += g
        k = a["Kernel_Name"].split("(")[0][-40:] + " -> " + b["Kernel_Name"].split("(")[0][-40:]
        gaps[k][0] += g; gaps[k][1] += 1
print(f"per step: span {span/nsteps/1e6:.3f} ms, kernel busy {busy/nsteps/1e6:.3f} ms, gaps {tot_gap/nsteps/1e6:.3f} ms, launches {len(seg)/nsteps:.0f}")
hist = collections.Counter()
prev_end = int(seg[0]["End_Timestamp"])
for a, b in zip(seg[:-1], seg[1:]):
    g = int(b["Start_Timestamp"]) - prev_end
    prev_end = max(prev_end, int(b["End_Timestamp"]))
    hist[min(max(g, 0) // 1000, 20)] += 1
print("gap histogram (us: count per step):", {k: round(v / nsteps, 1) for k, v in sorted(hist.items())})
for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {g/nsteps/1e3:8.1f} us/step  {n/nsteps:5.1f}x  avg {g/n/1e3:6.1f} us   {k}")
PY
find $o -name "*.csv" -delete; find $o -name "*.db" -delete
