mkdir -p gpurun_out/prof
bash scripts/gpu_final.sh
bash scripts/gpu_profile.sh r1e > gpurun_out/prof/r1e.log 2>&1; grep -E "rc=" gpurun_out/prof/r1e.log
bash scripts/gpu_profile_packnet.sh > gpurun_out/prof/pk.log 2>&1; grep -E "rc=" gpurun_out/prof/pk.log
