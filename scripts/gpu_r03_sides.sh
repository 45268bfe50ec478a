mkdir -p gpurun_out
one() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
{
for wl in sup_r50 mono_r18 sup_r18; do
echo "$wl side streams 1 / 2 / 3: $(one --workload $wl) $(one --workload $wl --const SIDE_STREAMS=2) $(one --workload $wl --const SIDE_STREAMS=3) | $(one --workload $wl) $(one --workload $wl --const SIDE_STREAMS=2)"
done
} > gpurun_out/r03z_sides.txt 2>&1
cat gpurun_out/r03z_sides.txt
