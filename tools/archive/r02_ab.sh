R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
for ws in 0 1; do for no in 0 1; do
  if [ $no = 1 ]; then export PIDM_NO_OVERLAP=1; else unset PIDM_NO_OVERLAP; fi
  echo -n "WGRAD_SPLIT=$ws NO_OVERLAP=$no: "; PIDM_WGRAD_SPLIT=$ws python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-alt 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done; done
