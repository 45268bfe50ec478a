mkdir -p gpurun_out
pk() { timeout -k 10 200 python bench.py --workload mono_packnet --no-cpu-baseline --profile-steps 0 --steps 8 --warmup 3 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
{
echo "packnet default: $(pk)"
echo "packnet GROUP_MAX_BYTES 32 / 64 / 256 / 512 MB: $(pk --const GROUP_MAX_BYTES=33554432) $(pk --const GROUP_MAX_BYTES=67108864) $(pk --const GROUP_MAX_BYTES=268435456) $(pk --const GROUP_MAX_BYTES=536870912)"
echo "packnet GROUP_BUDGET_BYTES 192 / 768 MB: $(pk --const GROUP_BUDGET_BYTES=201326592) $(pk --const GROUP_BUDGET_BYTES=805306368)"
echo "packnet DEFER_MAX_BYTES 0.5 / 8 MB: $(pk --const DEFER_MAX_BYTES=524288) $(pk --const DEFER_MAX_BYTES=8388608)"
echo "packnet (lag, group) (2,3) with blocks 1024 again: $(pk)   (3,2): $(pk --const JOIN_LAG=3 --const WGRAD_GROUP=2)  (2,1): $(pk --const JOIN_LAG=2 --const WGRAD_GROUP=1)"
} > gpurun_out/r03ah_packnet_sweep3.txt 2>&1
cat gpurun_out/r03ah_packnet_sweep3.txt
