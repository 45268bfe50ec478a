mkdir -p gpurun_out
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0 > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "failed $name"; tail -3 gpurun_out/ab.err; exit 1; }
  if grep -q "HSA_STATUS" gpurun_out/ab.err; then echo "fault $name"; exit 3; fi
  echo "$WL $name $(python -c "import json;d=json.load(open('gpurun_out/ab.json'));print(d['value'], d['ms_per_step'])")"
}
timeout -k 10 600 python -m pytest tests/test_gpu_nn.py -q -m gpu -k "group_norm" 2>&1 | tail -3
WL=mono_r18 run default SDE_X=0
WL=mono_r18 run default SDE_X=0


timeout -k 10 300 python -m pytest tests/test_gpu_models.py -q -m gpu -k "mono or packnet" 2>&1 | tail -3
