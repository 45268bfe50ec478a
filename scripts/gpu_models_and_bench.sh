mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_models.py -q -m gpu -s 2>&1 | tail -60 > gpurun_out/model_test.log; rc=$?
tail -8 gpurun_out/model_test.log
if [ $rc -ne 124 ] && [ $rc -ne 137 ]; then
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline > gpurun_out/bench_eager.json 2> gpurun_out/bench_eager.err; echo "eager rc=$?"; tail -c 1500 gpurun_out/bench_eager.json; tail -3 gpurun_out/bench_eager.err
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_graph.json 2> gpurun_out/bench_graph.err; echo "graph rc=$?"; tail -c 1500 gpurun_out/bench_graph.json; tail -3 gpurun_out/bench_graph.err
fi
