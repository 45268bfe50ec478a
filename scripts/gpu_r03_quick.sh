one() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
echo "sup_r50 default: $(one) $(one)"
echo "sup_r50 DEFER_MAX_BYTES 0.5 / 1 / 4 / 8 MB: $(one --const DEFER_MAX_BYTES=524288) $(one --const DEFER_MAX_BYTES=1048576) $(one --const DEFER_MAX_BYTES=4194304) $(one --const DEFER_MAX_BYTES=8388608)"
echo "sup_r50 GROUP_BUDGET_BYTES 128 / 256 / 768 MB: $(one --const GROUP_BUDGET_BYTES=134217728) $(one --const GROUP_BUDGET_BYTES=268435456) $(one --const GROUP_BUDGET_BYTES=805306368)"
echo "sup_r50 splitk off (opt 5=0): $(one --opt 5=0)   wgrad halo off (7=0): $(one --opt 7=0)   conv small off (8=0): $(one --opt 8=0)"
