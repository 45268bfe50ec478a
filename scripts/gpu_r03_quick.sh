timeout -k 10 900 python -m pytest tests/test_gpu_models.py -q -x 2>&1 | tail -2
timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sup_r50', d['ms_per_step'])"
timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 10 --force-overlap 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sup_r50 two-phase', d['ms_per_step'])"
