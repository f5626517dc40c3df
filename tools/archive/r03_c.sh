# round-3 call C: where does the graph mode lose time?  A/B of graph x side-stream overlap x launch stream, same box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r03c}; rm -rf $o; mkdir -p $o
run() { # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-alt 2>$o/$n.err | tail -1 > $o/$n.json
  python - $o/$n.json $n <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); l=d["launches"]
    print(f"{sys.argv[2]:28s} {d['value']:8.1f} samples/s  {d['ms_per_step']:7.3f} ms  host_enqueue {l['host_enqueue_ms_per_step']:7.3f} ms  graphs/step {l['graph_launches_per_step']}")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run graph_overlap        PIDM_GRAPH=1
run graph_nooverlap      PIDM_GRAPH=1 PIDM_NO_OVERLAP=1
run nograph_overlap      PIDM_GRAPH=0
run nograph_nooverlap    PIDM_GRAPH=0 PIDM_NO_OVERLAP=1
run graph_nooverlap_strm PIDM_GRAPH=1 PIDM_NO_OVERLAP=1 PIDM_BENCH_STREAM=1
run graph_overlap_strm   PIDM_GRAPH=1 PIDM_BENCH_STREAM=1
run nograph_overlap_strm PIDM_GRAPH=0 PIDM_BENCH_STREAM=1
run graph_nooverlap_eager PIDM_GRAPH=1 PIDM_NO_OVERLAP=1 PIDM_BENCH_EAGER=1
run nograph_overlap_eager PIDM_GRAPH=0 PIDM_BENCH_EAGER=1
ls $o | head -3
