mkdir -p gpurun_out
pk() { timeout -k 10 200 python bench.py --workload mono_packnet --no-cpu-baseline --profile-steps 0 --steps 8 --warmup 3 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
one() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
{
echo "packnet WGRAD_BLOCKS 1024 / 1536 / 2048 / 3072 / 4096: $(pk --opt 6=1024) $(pk --opt 6=1536) $(pk --opt 6=2048) $(pk --opt 6=3072) $(pk --opt 6=4096)"
echo "packnet 1024 + (2,2) / 2048 + (2,2): $(pk --opt 6=1024 --const JOIN_LAG=2 --const WGRAD_GROUP=2) $(pk --opt 6=2048 --const JOIN_LAG=2 --const WGRAD_GROUP=2)"
echo "sup_r50 WGRAD_BLOCKS 256 / 384 / 512 / 768: $(one) $(one --opt 6=384) $(one --opt 6=512) $(one --opt 6=768)"
echo "mono_r18 WGRAD_BLOCKS 256 / 512: $(one --workload mono_r18) $(one --workload mono_r18 --opt 6=512)"
} > gpurun_out/r03ah_packnet_sweep2.txt 2>&1
cat gpurun_out/r03ah_packnet_sweep2.txt
