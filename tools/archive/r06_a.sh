# round 6, first call: GPU suite + baseline bench line at HEAD + per-shape conv times (with the launcher's item / workgroup trace)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r06a}; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log; grep -E "passed|failed|rc=" $o/pytest.log | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 2>$o/bench.err | tail -1 > $o/bench.json
python -c "
import json; d=json.load(open('$o/bench.json')); print('bench', d['value'], d['ms_per_step'], d['step_flop_fraction'], d['north_star_b256']['value'], d['sampling_b1024']['value'], d['mechanics_b32']['value'])"
for b in 16 64 256; do
echo "#### batch $b" >> $o/conv_shapes.txt
PIDM_TRACE_CONV=1 timeout 300 python tools/bench_conv.py $b >> $o/conv_shapes.txt 2> $o/conv_trace_b$b.txt
done
grep -E "####|H=" $o/conv_shapes.txt
sort $o/conv_trace_b64.txt | uniq -c | sort -rn | head -60 > $o/conv_trace_b64_uniq.txt
