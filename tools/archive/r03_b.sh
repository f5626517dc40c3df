# round-3 call B: hipGraph replay on hardware - GPU tests, default bench, the same with PIDM_GRAPH=0 (A/B on one box), gap profiles
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r03b}; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log; tail -4 $o/pytest.log
timeout 600 python bench.py 2>$o/bench.err | tail -1 > $o/bench.json
PIDM_GRAPH=0 timeout 600 python bench.py --no-cpu-baseline 2>$o/bench_nograph.err | tail -1 > $o/bench_nograph.json
python - $o <<'PY'
import json,sys
o=sys.argv[1]
for f in ("bench.json","bench_nograph.json"):
    try:
        d=json.load(open(f"{o}/{f}"))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, d["value"], d["ms_per_step"], {k:(d.get(k) or {}).get("value") for k in ("fp32_mfma_only","eager_scalars","dropin_main_py","north_star_b256")}, d.get("launches"))
PY
tail -3 $o/bench.err
for g in 1 0; do
  d=$o/gaps_graph$g; mkdir -p $d
  (cd /tmp && PIDM_GRAPH=$g PIDM_NO_OVERLAP=1 timeout 400 rocprofv3 --kernel-trace --output-format csv -d $d/prof -o p -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-alt > $d/prof.log 2>&1)
  f=$(find $d -name "*kernel_trace.csv" | head -1)
  python - "$f" <<'PY' > $o/gaps_graph$g.txt 2>&1
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "clip_adam" in n]
lo, hi = idx[-9] + 1, idx[-1] + 1
seg = rows[lo:hi]; nsteps = 8
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
gaps = collections.defaultdict(lambda: [0, 0]); tot_gap = 0
prev_end = int(seg[0]["End_Timestamp"])
for a, b in zip(seg[:-1], seg[1:]):
    g = int(b["Start_Timestamp"]) - max(prev_end, int(a["End_Timestamp"]))
    prev_end = max(prev_end, int(a["End_Timestamp"]))
    if g > 0:
        tot_gap += g
        k = a["Kernel_Name"].split("(")[0][-40:] + " -> " + b["Kernel_Name"].split("(")[0][-40:]
        gaps[k][0] += g; gaps[k][1] += 1
print(f"per step: span {span/nsteps/1e6:.3f} ms, kernel busy {busy/nsteps/1e6:.3f} ms, gaps {tot_gap/nsteps/1e6:.3f} ms, launches {len(seg)/nsteps:.0f}")
for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {g/nsteps/1e3:8.1f} us/step  {n/nsteps:5.1f}x  avg {g/n/1e3:6.1f} us   {k}")
PY
  head -8 $o/gaps_graph$g.txt
  find $d -name "*.csv" -delete; find $d -name "*.db" -delete
done
ls $o
