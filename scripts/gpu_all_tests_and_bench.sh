mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -v -m gpu > gpurun_out/all_gpu_tests.log 2>&1; rc=$?
tail -8 gpurun_out/all_gpu_tests.log
if grep -q "HSA_STATUS_ERROR\|Aborted\|dumped core\|Fatal Python error" gpurun_out/all_gpu_tests.log; then echo "GPU fault in tests: stopping"; exit 3; fi
if [ $rc -eq 0 ] || [ $rc -eq 1 ]; then
  SDE_BENCH_LAYER_DUMP=gpurun_out/layers.csv timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 2 > gpurun_out/bench_graph.json 2> gpurun_out/bench_graph.err; echo "graph rc=$?"; tail -c 1700 gpurun_out/bench_graph.json; tail -2 gpurun_out/bench_graph.err
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --force-overlap > gpurun_out/bench_overlap.json 2> gpurun_out/bench_overlap.err; echo "overlap rc=$?"; tail -c 600 gpurun_out/bench_overlap.json; tail -2 gpurun_out/bench_overlap.err
  timeout -k 10 300 python bench.py --workload mono_r18 --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 2 > gpurun_out/bench_mono18.json 2> gpurun_out/bench_mono18.err; echo "mono18 rc=$?"; tail -c 2500 gpurun_out/bench_mono18.json; tail -2 gpurun_out/bench_mono18.err
fi
