mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -s 2>&1 | tail -40 > gpurun_out/all_gpu_tests.log; rc=$?
tail -12 gpurun_out/all_gpu_tests.log
if [ $rc -ne 124 ] && [ $rc -ne 137 ]; then
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-graph --no-cpu-baseline > gpurun_out/bench_eager.json 2> gpurun_out/bench_eager.err; echo "eager rc=$?"; tail -c 1800 gpurun_out/bench_eager.json; tail -2 gpurun_out/bench_eager.err
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 > gpurun_out/bench_graph.json 2> gpurun_out/bench_graph.err; echo "graph rc=$?"; tail -c 700 gpurun_out/bench_graph.json; tail -2 gpurun_out/bench_graph.err
fi
