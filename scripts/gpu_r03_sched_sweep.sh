# Re-tune the side-stream schedule after the BatchNorm-backward fusion shortened the data-gradient chain (one call: box-to-box variation is +-4 %)
mkdir -p gpurun_out
one() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 40 --warmup 8 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
echo "default: $(one) $(one)"
for lag in 1 2; do for grp in 3 4 6 8; do for first in 0 3 4; do
  echo "JOIN_LAG=$lag WGRAD_GROUP=$grp FIRST_GROUP=$first: $(one --const JOIN_LAG=$lag --const WGRAD_GROUP=$grp --const FIRST_GROUP=$first)"
done; done; done | tee gpurun_out/r03h_sched_sweep.txt
echo "default: $(one)"
