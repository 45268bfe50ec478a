one() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
echo "sup_r50 BNFA rows 256 / 384 / 512 / 768 / 1024: $(one) $(SDE_BNFA_MAX_ROWS=384 one) $(SDE_BNFA_MAX_ROWS=512 one) $(SDE_BNFA_MAX_ROWS=768 one) $(SDE_BNFA_MAX_ROWS=1024 one) | $(one)"
echo "mono_r18 256 / 768: $(one --workload mono_r18) $(SDE_BNFA_MAX_ROWS=768 one --workload mono_r18)"
