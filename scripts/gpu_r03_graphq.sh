# A/B of the HIP runtime's graph-queue knobs on the replayed step (no code change): how many parallel streams the graph executor uses
mkdir -p gpurun_out
one() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
{
for wl in mono_r18 sup_r50; do
  echo "$wl default: $(one --workload $wl) $(one --workload $wl)"
  for q in 1 2 3 4 6 8; do
    echo "$wl DEBUG_HIP_FORCE_GRAPH_QUEUES=$q: $(DEBUG_HIP_FORCE_GRAPH_QUEUES=$q one --workload $wl) $(DEBUG_HIP_FORCE_GRAPH_QUEUES=$q one --workload $wl)"
  done
done
echo "mono_r18 multi default: $(one --workload mono_r18 --opt photo_multi=1)"
for q in 3 4 6; do
  echo "mono_r18 multi QUEUES=$q: $(DEBUG_HIP_FORCE_GRAPH_QUEUES=$q one --workload mono_r18 --opt photo_multi=1)"
done
echo "mono_r18 DYNAMIC_QUEUES=0: $(DEBUG_HIP_DYNAMIC_QUEUES=0 one --workload mono_r18)  =1: $(DEBUG_HIP_DYNAMIC_QUEUES=1 one --workload mono_r18)"
} > gpurun_out/r03t_graph_queues.txt 2>&1
cat gpurun_out/r03t_graph_queues.txt
