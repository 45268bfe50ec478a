mkdir -p gpurun_out
for nb in 512 320 256 192 512; do
  SDE_WGRAD_BLOCKS=$nb timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0 > gpurun_out/wb_$nb.json 2> gpurun_out/wb_$nb.err || { echo "failed $nb"; tail -3 gpurun_out/wb_$nb.err; exit 1; }
  if grep -q "HSA_STATUS" gpurun_out/wb_$nb.err; then echo "fault $nb"; exit 3; fi
  echo "blocks=$nb $(python -c "import json;d=json.load(open('gpurun_out/wb_$nb.json'));print(d['value'], d['ms_per_step'])")"
done
