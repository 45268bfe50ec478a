mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/kf_tests.log; cat gpurun_out/kf_tests.log
