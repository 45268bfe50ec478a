mkdir -p gpurun_out
SDE_DIST_DEBUG=1 SDE_DIST_BACKEND=gloo timeout -k 10 200 python bench.py --gpus 2 --workload mono_r18 --steps 3 --warmup 2 --batch 4 --no-cpu-baseline --profile-steps 0 > gpurun_out/dp2_dbg.json 2> gpurun_out/dp2_dbg.err
grep "\[dist\]" gpurun_out/dp2_dbg.json gpurun_out/dp2_dbg.err | tail -4 | cut -c1-900
