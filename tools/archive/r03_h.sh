R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r03h}; rm -rf $o; mkdir -p $o
timeout 600 python -m pytest tests/test_kernels_conv.py -m gpu -x -q -k "split or bit_identical or 7x7 or groupnorm" 2>&1 | tail -2
for ws in 1 0; do echo "== PIDM_SPLIT_WS=$ws"; PIDM_SPLIT_WS=$ws python tools/bench_conv.py 64 2>/dev/null | grep -E "K=3|K=4|K=7|TOTAL" | cut -c1-150 | tee -a $o/conv_ws$ws.txt; done
for ws in 1 0; do
PIDM_SPLIT_WS=$ws timeout 600 python bench.py --no-cpu-baseline --no-alt --steps 40 2>$o/bench_ws$ws.err | tail -1 > $o/bench_ws$ws.json
python - $o/bench_ws$ws.json $ws <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("ws", sys.argv[2], d["value"], d["ms_per_step"], {k:r[k] for k in ("achieved","frac","frac_bf16_pipe","kernel_ms_per_step")}, r["split_form"], r["fwd_dgrad"])
PY
done
