# rocprofv3 kernel trace of the bench step (eager launches so that every kernel is a separate dispatch)
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1d -- python3 bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --profile-steps 0 > gpurun_out/prof/bench.json 2> gpurun_out/prof/bench.err
echo "rc=$?"; tail -c 600 gpurun_out/prof/bench.json
find gpurun_out/prof -name "*.csv" | head
f=$(ls gpurun_out/prof/*r1d_kernel_stats.csv | head -1)
head -40 "$f"
