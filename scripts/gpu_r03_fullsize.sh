mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_fullsize.py -q -x -s -k "${1:-packnet_1a or 200_steps}" > gpurun_out/r03f_fullsize.log 2>&1; rc=$?
grep -E "fp32 full size|loss: start|passed|failed|Error|assert" gpurun_out/r03f_fullsize.log | head -20
exit $rc
