mkdir -p gpurun_out
one() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 100 --warmup 20 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], (d.get('input_side') or {}).get('copy_stream') or '')"; }
{
for wl in mono_r18 mono_r50 sup_r50 sup_r18; do
echo "$wl resident / loader: $(one --workload $wl) | $(one --workload $wl --with-loader) | $(one --workload $wl --with-loader)"
done
} > gpurun_out/r03x_loader.txt 2>&1
cat gpurun_out/r03x_loader.txt
