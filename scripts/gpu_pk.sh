mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_nn.py -q -m gpu -k "conv3d" 2>&1 | tail -2
bash scripts/gpu_profile_packnet.sh 2>&1 | grep -E "rc=|conv3d|igemm_kernelIDF16bLi64ELi64ELi2ELi2ELi0ELb1|wgrad_kernelIDF16bLi64ELi128ELi1ELi4ELi0ELb1" | cut -c1-200
bash scripts/gpu_profile.sh r1e > gpurun_out/prof/r1e.log 2>&1; tail -60 gpurun_out/prof/r1e.log | cut -c1-220
