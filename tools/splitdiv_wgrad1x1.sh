mkdir -p gpurun_out/r06_i; o=$PWD/gpurun_out/r06_i; R=$PWD
cd /tmp && export TMPDIR=/tmp
for m in 1 2 4; do
  PIDM_WGRAD1X1_SPLIT_DIV=$m PIDM_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/tm_$m -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-roofline > $o/tm_$m.log 2>&1
  echo "split divisor $m"; grep -h "conv_wgrad_1x1_split_multi\|reduce_multi" $o/tm_$m/p_kernel_stats.csv | sed 's/(.*)"//' | cut -d, -f1-4
done
