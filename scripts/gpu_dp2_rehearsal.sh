# two ranks sharing the ONE GPU of the box over gloo: exercises the N>1 control path of bench.py / HipTrainer
# (self-launch of the ranks, initial broadcast, two-phase backward with graph A / graph B, async bucketed all-reduce, 1/world in Adam).
# Invoked as the driver invokes the N=1 case -- plain `python bench.py --gpus 2`, no launcher.  Not a perf number.
mkdir -p gpurun_out
SDE_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 5 --warmup 2 --batch 4 --no-cpu-baseline --profile-steps 0 \
   > gpurun_out/dp2_rehearsal.json 2> gpurun_out/dp2_rehearsal.err
echo "rc=$?"; tail -c 800 gpurun_out/dp2_rehearsal.json; tail -5 gpurun_out/dp2_rehearsal.err
