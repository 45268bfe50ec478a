mkdir -p gpurun_out
SDE_BENCH_LAYER_DUMP=gpurun_out/layers.csv timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline --profile-steps 1 > gpurun_out/bench_dump.json 2> gpurun_out/bench_dump.err; echo rc=$?
wc -l gpurun_out/layers.csv
