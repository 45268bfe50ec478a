mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_nn.py tests/test_gpu_models.py -q -m gpu 2>&1 | tail -25 > gpurun_out/nn_models.log; rc=$?
tail -6 gpurun_out/nn_models.log
if [ $rc -ne 124 ] && [ $rc -ne 137 ]; then
  SDE_BENCH_LAYER_DUMP=gpurun_out/layers.csv timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 2 > gpurun_out/bench_graph.json 2> gpurun_out/bench_graph.err; echo "graph rc=$?"; tail -c 1700 gpurun_out/bench_graph.json; tail -2 gpurun_out/bench_graph.err
fi
