mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_models.py -q -m gpu 2>&1 | tail -6 > gpurun_out/kf_tests.log; cat gpurun_out/kf_tests.log
