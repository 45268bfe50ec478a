mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_photometric.py tests/test_gpu_models.py -q -m gpu -k "photo or mono or pose" 2>&1 | tail -3
bash scripts/gpu_profile_packnet.sh
WL=mono_r18; timeout -k 10 200 python bench.py --workload mono_r18 --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('mono_r18', d['value'], d['ms_per_step'])"
