mkdir -p gpurun_out
B=$PWD/simpledepthestimation_amd/libsde_hip_b.so
SDE_HIP_LIB=$B timeout -k 10 300 python -m pytest tests/test_gpu_nn.py -q -m gpu -x 2>&1 | tail -3
for v in A B; do
  if [ $v = B ]; then export SDE_HIP_LIB=$B; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 2 > gpurun_out/bench_$v.json 2> gpurun_out/bench_$v.err; echo "$v rc=$?"
  python - <<PY
import json
d=json.loads(open('gpurun_out/bench_$v.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$v', d['value'], 'img/s', d['ms_per_step'], 'ms; gemm ms', r['gemm_ms_per_step'], {k:(v['ms_per_step'],v['tflops']) for k,v in r['families'].items()})
PY
done
