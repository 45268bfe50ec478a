mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_nn.py -q -x -k "cu_reserve or conv_batchnorm or conv_fwd_bwd" 2>&1 | tail -3
one() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
echo "sup_r50 reserve 0 / 16 / 32 / 64: $(one) $(one --opt 11=16) $(one --opt 11=32) $(one --opt 11=64)   force-overlap 0 / 32: $(one --force-overlap) $(one --force-overlap --opt 11=32)" | tee gpurun_out/r03ae_cu_reserve.txt
