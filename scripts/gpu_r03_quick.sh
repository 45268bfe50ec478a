mkdir -p gpurun_out/final
timeout -k 10 600 python -m pytest tests/test_gpu_nn.py -q -x -k "conv_fwd_bwd or cu_reserve" 2>&1 | tail -2
pk() { timeout -k 10 200 python bench.py --workload mono_packnet --no-cpu-baseline --profile-steps 0 --steps 8 --warmup 3 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
echo "packnet: $(pk) $(pk)  sup_r50: $(timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
timeout -k 10 300 python bench.py --workload mono_packnet --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/final/bench_mono_packnet.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/final/bench_mono_packnet.json').read().strip().splitlines()[-1]); print('packnet line', d['value'], d['ms_per_step'])"
