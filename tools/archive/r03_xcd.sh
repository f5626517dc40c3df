R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
export BENCH_CONV_SHAPES="16,128,0,768,1,1,0,0;16,256,0,128,1,1,0,0;8,256,0,768,1,1,0,0;8,512,0,256,1,1,0,0;32,128,0,64,1,1,0,0"
for x in 0 1; do echo "== B=64 XCD=$x"; PIDM_WGRAD_XCD=$x python tools/bench_conv.py 64 2>/dev/null | grep "K=1" | cut -c95-140; done
for x in 0 1; do echo "== B=256 XCD=$x"; PIDM_WGRAD_XCD=$x python tools/bench_conv.py 256 2>/dev/null | grep "K=1" | cut -c95-140; done
export BENCH_CONV_SHAPES="64,128,0,768,1,1,0,0;64,256,0,128,1,1,0,0;32,256,0,768,1,1,0,0;32,512,0,256,1,1,0,0;16,512,0,768,1,1,0,0;16,1024,0,512,1,1,0,0;8,1024,0,768,1,1,0,0"
for x in 0 1; do echo "== mechanics B=32 XCD=$x"; PIDM_WGRAD_XCD=$x python tools/bench_conv.py 32 2>/dev/null | grep "K=1" | cut -c95-140; done
