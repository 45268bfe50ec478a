# round-end measurement: full GPU suite, default bench line (with cpu_baseline), MonoDepth2 / PackNet lines, 2-rank rehearsal
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -v -m gpu > gpurun_out/gpu_tests.log 2>&1; rc=$?
grep -E "passed|failed|error" gpurun_out/gpu_tests.log | tail -3
if grep -q "HSA_STATUS_ERROR\|Aborted\|dumped core\|Fatal Python error" gpurun_out/gpu_tests.log; then echo "GPU fault in tests"; exit 3; fi
grep -E "^E  |FAILED" gpurun_out/gpu_tests.log | head -20
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout -k 10 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; cat gpurun_out/bench_default.json | cut -c1-3000
timeout -k 10 300 python bench.py --workload mono_r18 --no-cpu-baseline > gpurun_out/bench_mono18.json 2> gpurun_out/bench_mono18.err; echo "mono18 rc=$?"; cut -c1-400 gpurun_out/bench_mono18.json
timeout -k 10 300 python bench.py --workload mono_r50 --no-cpu-baseline --profile-steps 0 > gpurun_out/bench_mono50.json 2> gpurun_out/bench_mono50.err; echo "mono50 rc=$?"; cut -c1-400 gpurun_out/bench_mono50.json
timeout -k 10 500 python bench.py --workload mono_packnet --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 1 > gpurun_out/bench_packnet.json 2> gpurun_out/bench_packnet.err; echo "packnet rc=$?"; cut -c1-600 gpurun_out/bench_packnet.json; tail -2 gpurun_out/bench_packnet.err
