one() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
echo "sup_r50 resident: $(one) $(one)"
echo "sup_r50 loader: $(one --with-loader) $(one --with-loader) $(one --with-loader) $(one --with-loader)"
echo "sup_r50 loader, 40 warm-up steps: $(one --with-loader --warmup 40) $(one --with-loader --warmup 40)"
echo "mono_r18 resident: $(one --workload mono_r18); loader: $(one --workload mono_r18 --with-loader) $(one --workload mono_r18 --with-loader) ; warm-up 40: $(one --workload mono_r18 --with-loader --warmup 40)"
echo "mono_r50 resident: $(one --workload mono_r50); loader: $(one --workload mono_r50 --with-loader) $(one --workload mono_r50 --with-loader)"
