mkdir -p gpurun_out
run() { env "$@" timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], 'ms', d['value'], 'img/s')"; }
for round in 1 2 3; do
  run SDE_NO_HALO=1
  run SDE_NO_HALO=0 SDE_HALO_MIN_N=0
  run SDE_NO_HALO=0 SDE_HALO_MIN_N=64
  run SDE_NO_HALO=0 SDE_HALO_MIN_N=64 SDE_WGRAD_SIDE_STREAM=0
  run SDE_NO_HALO=1 SDE_WGRAD_SIDE_STREAM=0
done
