R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
for v in "" "PIDM_NO_BN_EPILOGUE=1" "PIDM_NO_GN_EPILOGUE=1" ""; do
env $v timeout 300 python bench.py --no-cpu-baseline --no-alt --no-roofline --steps 40 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
done
