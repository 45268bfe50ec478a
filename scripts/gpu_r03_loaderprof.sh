mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/prof -o r03d_ld -- python3 scripts/loader_host_profile.py sup_r50 short > gpurun_out/r03d_ld.txt 2>&1
echo rc=$?; tail -5 gpurun_out/r03d_ld.txt
ls gpurun_out/prof | grep r03d_ld
f=$(ls gpurun_out/prof/*r03d_ld_kernel_stats.csv | head -1); grep -E "image_prep|copyBuffer|Name" "$f" | cut -c1-200
f=$(ls gpurun_out/prof/*r03d_ld_memory_copy_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cat "$f" | cut -c1-200
