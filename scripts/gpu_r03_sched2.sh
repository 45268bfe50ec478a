mkdir -p gpurun_out
one() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
{
echo "sup_r50 (1,6,0) default: $(one) $(one)"
for cfg in "1 4 0" "1 5 0" "1 7 0" "1 8 0" "1 6 3" "1 6 4" "2 6 0" "2 4 0" "1 5 3" "1 8 4"; do set -- $cfg; echo "sup_r50 ($1,$2,$3): $(one --const JOIN_LAG=$1 --const WGRAD_GROUP=$2 --const FIRST_GROUP=$3)"; done
echo "mono_r18 (1,6,3) default: $(one --workload mono_r18)"
for cfg in "1 5 3" "1 7 3" "1 6 2" "1 6 4" "1 4 2" "1 8 3"; do set -- $cfg; echo "mono_r18 ($1,$2,$3): $(one --workload mono_r18 --const JOIN_LAG=$1 --const WGRAD_GROUP=$2 --const FIRST_GROUP=$3)"; done
} > gpurun_out/r03ai_sched.txt 2>&1
cat gpurun_out/r03ai_sched.txt
