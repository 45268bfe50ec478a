mkdir -p gpurun_out
for cfg in "128 256" "64 256" "128 384" "128 192" "64 384" "128 256"; do
  set -- $cfg
  SDE_HALO_BN_MAX=$1 SDE_WGRAD_BLOCKS=$2 timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0 > gpurun_out/bt.json 2> gpurun_out/bt.err || { echo "failed $cfg"; tail -3 gpurun_out/bt.err; exit 1; }
  if grep -q "HSA_STATUS" gpurun_out/bt.err; then echo "fault $cfg"; exit 3; fi
  echo "halo_bn_max=$1 wgrad_blocks=$2 $(python -c "import json;d=json.load(open('gpurun_out/bt.json'));print(d['value'], d['ms_per_step'])")"
done
