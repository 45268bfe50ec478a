mkdir -p gpurun_out
: > gpurun_out/ablate.log
for d in 0 1 2 3 4 8 12 15; do
  echo "=== SDE_CONV_DEBUG=$d ===" >> gpurun_out/ablate.log
  SDE_CONV_DEBUG=$d timeout -k 10 200 python scripts/microbench_conv.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/ablate.log
done
cat gpurun_out/ablate.log
