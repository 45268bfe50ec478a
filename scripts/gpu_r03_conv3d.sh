mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_nn.py -q -x -k "conv3d" > gpurun_out/r03i_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r03i_tests.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert|FAILED" gpurun_out/r03i_tests.log | head -20; exit $rc; fi

timeout -k 10 250 python scripts/microbench_conv3d.py 2>&1 | grep -v amdgpu.ids | tail -12
