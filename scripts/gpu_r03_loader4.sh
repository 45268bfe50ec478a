one() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for wl in sup_r50 mono_r18 mono_r50; do
echo "$wl resident: $(one --workload $wl); in place: $(one --workload $wl --with-loader) $(one --workload $wl --with-loader); copy stream: $(one --workload $wl --with-loader --aug-on-copy-stream) $(one --workload $wl --with-loader --aug-on-copy-stream)"
done
