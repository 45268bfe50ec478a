mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_nn.py tests/test_gpu_models.py -q -m gpu 2>&1 | tail -4
for ss in 1 0; do
  SDE_WGRAD_SIDE_STREAM=$ss timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 > gpurun_out/bench_ss$ss.json 2> gpurun_out/bench_ss$ss.err; echo "side_stream=$ss rc=$?"; tail -c 330 gpurun_out/bench_ss$ss.json | head -c 200; echo; tail -2 gpurun_out/bench_ss$ss.err | cut -c1-300
done
bash scripts/gpu_dp2_rehearsal.sh
