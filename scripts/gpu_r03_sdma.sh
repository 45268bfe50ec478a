one() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
echo "resident: $(one) $(one)"
echo "loader default env: $(one --with-loader) $(one --with-loader) $(one --with-loader)"
echo "loader HSA_ENABLE_SDMA=0: $(HSA_ENABLE_SDMA=0 one --with-loader) $(HSA_ENABLE_SDMA=0 one --with-loader)"
echo "loader HSA_ENABLE_SDMA=1: $(HSA_ENABLE_SDMA=1 one --with-loader) $(HSA_ENABLE_SDMA=1 one --with-loader)"
echo "loader GPU_MAX_HW_QUEUES=8: $(GPU_MAX_HW_QUEUES=8 one --with-loader) $(GPU_MAX_HW_QUEUES=8 one --with-loader)"
echo "mono_r18 resident: $(one --workload mono_r18); loader: $(one --workload mono_r18 --with-loader) $(one --workload mono_r18 --with-loader); SDMA=0: $(HSA_ENABLE_SDMA=0 one --workload mono_r18 --with-loader); HWQ=8: $(GPU_MAX_HW_QUEUES=8 one --workload mono_r18 --with-loader)"
