R=${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp PYTHONPATH=$R; out=$R/gpurun_out/pa; rm -rf $out; mkdir -p $out
true
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python $R/tools/bench_attn.py 64 64 > $out/log.txt 2>&1)
python - "$out/p_kernel_trace.csv" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name'].replace('void pidm::','').replace('pidm::','').split('(')[0]
    if not n.startswith('la_'): continue
    d[(n, r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''), r.get('Grid_Size_Y',''))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    v=sorted(v); print(f"{k[0]:34s} grid=({k[1]},{k[2]}) n={len(v):3d} med={v[len(v)//2]:8.1f}us")
PY
