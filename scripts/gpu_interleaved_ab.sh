# interleaved A/B of end-to-end step time: $1 = env var name, values 0/1 alternate, 4 rounds each
mkdir -p gpurun_out
VAR=${1:-SDE_NO_HALO}
for round in 1 2 3 4; do
  for v in 0 1; do
    env $VAR=$v timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v round $round', d['ms_per_step'], 'ms', d['value'], 'img/s')"
  done
done
