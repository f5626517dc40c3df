# backward pass: graph replay vs launch by launch with the side-stream overlap (forward replayed in both), all bench legs, same box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
for cfg in "PIDM_GRAPH=1" "PIDM_GRAPH_BWD=0" "PIDM_GRAPH=0" "PIDM_GRAPH=1" "PIDM_GRAPH_BWD=0"; do
  env $cfg timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', 'headline', d['value'], d['ms_per_step'], 'eager_scalars', d['eager_scalars']['value'], 'dropin', d['dropin_main_py']['value'], 'b256', d['north_star_b256']['value'], d['north_star_b256']['ms_per_step'])"
done
for cfg in "PIDM_GRAPH=1" "PIDM_GRAPH_BWD=0"; do
  env $cfg timeout 600 python bench.py --workload mechanics --no-cpu-baseline --no-roofline --steps 20 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', 'mechanics', d['value'], d['ms_per_step'], 'eager_scalars', d['eager_scalars']['value'], 'dropin', d['dropin_main_py']['value'])"
done
