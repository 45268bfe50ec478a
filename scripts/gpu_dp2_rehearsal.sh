# two ranks sharing the ONE GPU of the box over gloo: exercises the N>1 control path of bench.py / HipTrainer
# (self-launch of the ranks, initial broadcast, two-phase backward with graph A / graph B, async bucketed all-reduce, 1/world in Adam).
# Invoked as the driver invokes the N=1 case -- plain `python bench.py --gpus 2`, no launcher -- and through torch.distributed.run.  Not a perf number.
mkdir -p gpurun_out
for wl in sup_r50 mono_r18; do
SDE_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --workload $wl --steps 5 --warmup 2 --batch 4 --no-cpu-baseline --profile-steps 0 \
   > gpurun_out/dp2_rehearsal_$wl.json 2> gpurun_out/dp2_rehearsal_$wl.err
echo "$wl rc=$?"; tail -c 700 gpurun_out/dp2_rehearsal_$wl.json; tail -3 gpurun_out/dp2_rehearsal_$wl.err
done
SDE_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --batch 4 --no-cpu-baseline --profile-steps 0 \
   > gpurun_out/dp2_rehearsal_torchrun.json 2> gpurun_out/dp2_rehearsal_torchrun.err
echo "torchrun rc=$?"; tail -c 500 gpurun_out/dp2_rehearsal_torchrun.json; tail -3 gpurun_out/dp2_rehearsal_torchrun.err
