mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -v -m gpu -x > gpurun_out/gpu_tests.log 2>&1; rc=$?
grep -E "passed|failed|error" gpurun_out/gpu_tests.log | tail -5
if grep -q "HSA_STATUS_ERROR\|Aborted\|dumped core\|Fatal Python error" gpurun_out/gpu_tests.log; then echo "GPU fault in tests"; grep -E "^tests/.*(PASSED|FAILED)|::" gpurun_out/gpu_tests.log | tail -3; exit 3; fi
grep -E "FAILED|Error" gpurun_out/gpu_tests.log | head -20
exit $rc
