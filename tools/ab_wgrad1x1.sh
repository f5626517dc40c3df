mkdir -p gpurun_out/r06_i; o=$PWD/gpurun_out/r06_i
python -m pytest tests/test_kernels_conv.py tests/test_unet_engine.py -m gpu -x -q 2>&1 | tail -4
python -m pytest tests/test_kernels_conv.py -m gpu -q -s -k "1x1_split_wgrad" 2>&1 | grep "max error"
for rep in 1 2; do
for v in 1 0; do
  PIDM_WGRAD1X1_SPLIT=$v python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-alt --no-roofline > $o/ab_b64_$v.json 2>/dev/null
  PIDM_WGRAD1X1_SPLIT=$v python bench.py --batch 256 --steps 10 --warmup 4 --no-cpu-baseline --no-alt --no-roofline > $o/ab_b256_$v.json 2>/dev/null
  PIDM_WGRAD1X1_SPLIT=$v python bench.py --workload mechanics --steps 10 --warmup 4 --no-cpu-baseline --no-alt --no-roofline > $o/ab_mech_$v.json 2>/dev/null
  python - <<P
import json
for n in ("b64","b256","mech"):
    d=json.loads(open("$o/ab_%s_$v.json"%n).read().strip().splitlines()[-1])
    print("split=$v",n,d["value"],d["ms_per_step"])
P
done
done
