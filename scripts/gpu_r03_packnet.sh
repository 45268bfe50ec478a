# Round 3: PackNet01 with native data movement -- its tests, the bench lines (bf16, fp16 + loss scaling) and a rocprofv3 kernel-stats pass of the workload.
mkdir -p gpurun_out/prof
timeout -k 10 900 python -m pytest tests/test_gpu_nn.py tests/test_gpu_models.py -q -x -k "space_to_depth or concat or inv_depth or group_norm or packnet or conv3d or PackNet" > gpurun_out/r03c_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r03c_tests.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert|FAILED" gpurun_out/r03c_tests.log | head -30; exit $rc; fi
timeout -k 10 300 python bench.py --workload mono_packnet --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r03c_bench_mono_packnet.json 2> gpurun_out/r03c_err.log || { tail -5 gpurun_out/r03c_err.log; exit 5; }
cut -c1-400 gpurun_out/r03c_bench_mono_packnet.json
timeout -k 10 300 python bench.py --workload mono_packnet --dtype fp16 --no-cpu-baseline --profile-steps 0 --steps 10 --warmup 3 > gpurun_out/r03c_bench_mono_packnet_fp16.json 2> gpurun_out/r03c_err.log || { tail -5 gpurun_out/r03c_err.log; exit 5; }
cut -c1-300 gpurun_out/r03c_bench_mono_packnet_fp16.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r03c_pk -- python3 bench.py --workload mono_packnet --steps 3 --warmup 2 --no-graph --no-cpu-baseline --profile-steps 0 > gpurun_out/prof/r03c_pk_bench.json 2> gpurun_out/prof/r03c_pk_bench.err
echo "trace rc=$?"
f=$(ls gpurun_out/prof/*r03c_pk_kernel_stats.csv | head -1); cp "$f" gpurun_out/r03c_packnet_kernel_stats.csv; head -30 "$f" | cut -c1-180
