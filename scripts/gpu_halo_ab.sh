mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_nn.py -q -m gpu -x -k "halo" 2>&1 | tail -15
rc=$?
timeout -k 10 600 python -m pytest tests/test_gpu_nn.py tests/test_gpu_models.py -q -m gpu 2>&1 | tail -4
for nh in 1 0; do
  SDE_BENCH_LAYER_DUMP=gpurun_out/layers_nh$nh.csv SDE_NO_HALO=$nh timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 2 > gpurun_out/bench_nh$nh.json 2> gpurun_out/bench_nh$nh.err; echo "no_halo=$nh rc=$?"
  python - <<PY
import json
d=json.loads(open('gpurun_out/bench_nh$nh.json').read().strip().splitlines()[-1])
r=d['roofline']
print('no_halo=$nh', d['value'], 'img/s', d['ms_per_step'], 'ms; gemm ms', r['gemm_ms_per_step'], {k:(v['ms_per_step'],v['tflops'],v['launches_per_step']) for k,v in r['families'].items()})
PY
  tail -2 gpurun_out/bench_nh$nh.err | cut -c1-300
done
