mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o pk -- python3 bench.py --workload mono_packnet --batch 4 --steps 2 --warmup 1 --no-graph --no-cpu-baseline --profile-steps 0 > gpurun_out/prof/pk_bench.json 2> gpurun_out/prof/pk_bench.err
echo "rc=$?"; tail -c 300 gpurun_out/prof/pk_bench.json
f=$(ls gpurun_out/prof/*pk_kernel_stats.csv | head -1); head -40 "$f" | cut -c1-220
