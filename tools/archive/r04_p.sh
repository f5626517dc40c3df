# round 4, call P: what precedes each __amd_rocclr_copyBuffer of a Darcy step (kernel trace in order)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/${1:-r04p}; mkdir -p $O
(cd /tmp && PIDM_NO_OVERLAP=1 PIDM_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o p -- python $R/bench.py --batch 64 --steps 3 --warmup 2 --no-cpu-baseline --no-alt --no-roofline > $O/prof.log 2>&1)
python - $O <<'PY'
import csv,glob,sys,re
f=glob.glob(sys.argv[1]+'/prof/**/p_kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[re.sub(r'\(.*','',r['Kernel_Name']).replace('void ','').replace('pidm::','')[:60] for r in rows]
# last step: from the last qsample_kernel to the end
idx=[i for i,n in enumerate(names) if n.startswith('qsample')]
s=idx[-1]
e=len(names)
out=[]
for i in range(max(0,s-12),e):
    if 'copyBuffer' in names[i] or 'fillBuffer' in names[i] or 'at::' in names[i] or 'elementwise' in names[i]:
        out.append(f"{i-s:5d} {names[i]:40s} dur {int(rows[i]['End_Timestamp'])-int(rows[i]['Start_Timestamp'])} ns | prev: {names[i-1]} | next: {names[i+1] if i+1<e else ''}")
print(f"step = {e-s} dispatches after the last qsample")
print("\n".join(out))
PY
find $O -name '*.db' -delete
