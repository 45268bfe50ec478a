mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof -o r03y_ld -- python3 bench.py --workload mono_r18 --with-loader --no-cpu-baseline --profile-steps 0 --steps 20 --warmup 10 > gpurun_out/r03y_ld.txt 2>&1
echo rc=$?; tail -2 gpurun_out/r03y_ld.txt | cut -c1-300
ls -la gpurun_out/prof | grep r03y_ld
