# cold-start cost per launch: the same kernels back to back (bench_conv: code and weights warm) against their in-step durations (kernel trace of the step)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r06m}; rm -rf $o; mkdir -p $o
export BENCH_CONV_SHAPES="64,32,0,32,3,1,1,0;32,64,0,64,3,1,1,0;16,128,0,128,3,1,1,0;8,256,0,256,3,1,1,0;16,128,0,384,1,1,0,0;8,256,0,768,1,1,0,0"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $o/warm -o p -- python $R/tools/bench_conv.py 64 > $o/warm.log 2>&1)
(cd /tmp && PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $o/step -o p -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-alt --no-roofline > $o/step.log 2>&1)
find $o -name '*.db' -delete; find $o -name '*agent_info.csv' -delete
python - $o <<'PY'
import csv, sys, glob, collections
o = sys.argv[1]
def load(d):
    f = glob.glob(f"{o}/{d}/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"].split("(")[0]
        key = (name, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0))))
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return agg
w, s = load("warm"), load("step")
print(f"{'kernel':60s} {'grid':>8s} {'warm n':>6s} {'warm us':>8s} {'step n':>6s} {'step us':>8s} {'delta':>7s}")
for k in sorted(w, key=lambda k: -sum(w[k])):
    if k in s and len(w[k]) >= 5:
        a = sorted(w[k])[len(w[k]) // 2]; b = sorted(s[k])[len(s[k]) // 2]
        print(f"{k[0][:60]:60s} {k[1]:8d} {len(w[k]):6d} {a:8.2f} {len(s[k]):6d} {b:8.2f} {b - a:+7.2f}")
PY
find $o -name '*kernel_trace.csv' -delete
