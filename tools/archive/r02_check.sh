# round-2 first GPU pass (one gpurun call): full -m gpu suite, the three bench workloads, the DP driver on one GPU, the
# overlapped gradient exchange with two ranks sharing the GPU (gloo), kernel statistics of the default bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r02a}; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
timeout 400 python bench.py 2>$o/bench.err | tail -1 > $o/bench.json
timeout 300 python bench.py --ema --no-cpu-baseline --no-roofline 2>>$o/bench.err | tail -1 > $o/bench_ema.json
timeout 400 python bench.py --workload mechanics --steps 10 --warmup 3 2>$o/bench_mech.err | tail -1 > $o/bench_mech.json
timeout 400 python bench.py --workload sampling --steps 20 --warmup 5 2>$o/bench_samp.err | tail -1 > $o/bench_samp.json
timeout 300 python main_dp.py --gov-eqs darcy --iterations 12 --synthetic --ema-start 3 --log-freq 4 --name dp_smoke > $o/main_dp.log 2>&1; echo "main_dp rc=$?" >> $o/main_dp.log
PIDM_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $o/bench_share2.log 2>&1; echo "share2 rc=$?" >> $o/bench_share2.log
(cd /tmp && PIDM_NO_OVERLAP=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/prof.log 2>&1)
tail -5 $o/pytest.log; cut -c1-400 $o/bench.json; cut -c1-200 $o/bench_ema.json; cut -c1-300 $o/bench_mech.json; cut -c1-300 $o/bench_samp.json; tail -4 $o/main_dp.log; tail -3 $o/bench_share2.log | cut -c1-300
