mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_pgemm.py tests/test_gpu_nn.py tests/test_gpu_fullsize.py -q -x -k "not 200_steps and not packnet_1a and not monodepth2_resnet50" > gpurun_out/r03m_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r03m_tests.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert|FAILED" gpurun_out/r03m_tests.log | head -20; exit $rc; fi
one() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 40 --warmup 8 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
echo "sup_r50 auto tile: $(one) ; forced 64x64: $(one --opt 4=64064) ; auto: $(one) ; 64x64: $(one --opt 4=64064)"
echo "mono_r18 auto: $(one --workload mono_r18) ; 64x64: $(one --workload mono_r18 --opt 4=64064)"
echo "mono_r50 auto: $(one --workload mono_r50) ; 64x64: $(one --workload mono_r50 --opt 4=64064)"
echo "sup_r18 auto: $(one --workload sup_r18) ; 64x64: $(one --workload sup_r18 --opt 4=64064)"
