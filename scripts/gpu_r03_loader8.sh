mkdir -p gpurun_out
{
for wl in mono_r18 sup_r50; do
for k in 0 1 2 3 4 5; do echo "$wl SKIP_STREAMS=$k: $(SKIP_STREAMS=$k timeout -k 10 60 python scripts/loader_gap_probe.py $wl normal 2>&1 | tail -1)"; done
done
} > gpurun_out/r03x_gap_probe.txt
cat gpurun_out/r03x_gap_probe.txt
