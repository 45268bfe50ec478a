mkdir -p gpurun_out
for k in 512 1024 2304 4608 1000000 0; do
  SDE_SHORTK=$k timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0 > gpurun_out/sk_$k.json 2> gpurun_out/sk_$k.err || { echo "failed $k"; tail -3 gpurun_out/sk_$k.err; exit 1; }
  if grep -q "HSA_STATUS" gpurun_out/sk_$k.err; then echo "fault $k"; exit 3; fi
  echo "shortk=$k $(python -c "import json;d=json.load(open('gpurun_out/sk_$k.json'));print(d['value'], d['ms_per_step'])")"
done
