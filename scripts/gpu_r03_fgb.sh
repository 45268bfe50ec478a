mkdir -p gpurun_out
one() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
{
for wl in sup_r50 mono_r18; do
echo "$wl plain early tail on / off: $(one --workload $wl) $(one --workload $wl --opt early_tail=0) $(one --workload $wl) $(one --workload $wl --opt early_tail=0)"
echo "$wl two-phase early tail on / off: $(one --workload $wl --force-overlap) $(one --workload $wl --force-overlap --opt early_tail=0)"
echo "$wl two-phase, first group of phase B 1 / 2 / 3 / 4 / 5 / 33: $(one --workload $wl --force-overlap --const FIRST_GROUP_B=1) $(one --workload $wl --force-overlap --const FIRST_GROUP_B=2) $(one --workload $wl --force-overlap --const FIRST_GROUP_B=3) $(one --workload $wl --force-overlap --const FIRST_GROUP_B=4) $(one --workload $wl --force-overlap --const FIRST_GROUP_B=5) $(one --workload $wl --force-overlap --const FIRST_GROUP_B=33)"
done
} > gpurun_out/r03ag_fgb.txt 2>&1
cat gpurun_out/r03ag_fgb.txt
