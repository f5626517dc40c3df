R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R; o=$R/gpurun_out/${1:-r06d}; rm -rf $o; mkdir -p $o
for sh in "8 256 256 64" "16 128 128 64" "8 256 256 256"; do
for ws in 0 1; do
echo "######## PIDM_SPLIT_WS=$ws shape $sh" >> $o/trace.txt
PIDM_SPLIT_WS=$ws timeout 120 python tools/conv_trace.py $sh >> $o/trace.txt 2>&1
done; done
cat $o/trace.txt
