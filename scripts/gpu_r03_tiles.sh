mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_pgemm.py -q -x > gpurun_out/r03l_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r03l_tests.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert|FAILED" gpurun_out/r03l_tests.log | head -20; exit $rc; fi
timeout -k 10 500 python scripts/microbench_gemm.py r50 pg3 pg3+3 pg3m pg3l 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03l_gemm_microbench.txt | tail -70
