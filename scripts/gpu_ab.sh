mkdir -p gpurun_out
echo "=== A (2-stage) ===" > gpurun_out/ab.log
timeout -k 10 300 python scripts/microbench_conv.py >> gpurun_out/ab.log 2>&1
echo "=== B (3-stage) ===" >> gpurun_out/ab.log
SDE_HIP_LIB=$PWD/simpledepthestimation_amd/libsde_hip_b.so timeout -k 10 300 python scripts/microbench_conv.py >> gpurun_out/ab.log 2>&1
SDE_HIP_LIB=$PWD/simpledepthestimation_amd/libsde_hip_b.so timeout -k 10 300 python -m pytest tests/test_gpu_nn.py -q -m gpu -x 2>&1 | tail -3 >> gpurun_out/ab.log
cat gpurun_out/ab.log
