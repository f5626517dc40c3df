for rep in 1 2 3; do
 for lib in new old; do
  export PIDM_LIBRARY=; [ $lib = old ] && export PIDM_LIBRARY=$PWD/ab_old/libpidm_hip.so; [ $lib = new ] && unset PIDM_LIBRARY
  for b in 64 16 256; do
   st=40; [ $b = 256 ] && st=12
   python bench.py --batch $b --steps $st --warmup 10 --no-cpu-baseline --no-alt --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib b$b', d['ms_per_step'])"
  done
  python bench.py --workload mechanics --steps 10 --warmup 4 --no-cpu-baseline --no-alt --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib mech', d['ms_per_step'])"
  python bench.py --workload sampling --steps 10 --warmup 4 --no-cpu-baseline --no-alt --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib samp', d['ms_per_step'])"
 done
done
