mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_models.py -q -x -k "phase or graph_replay or supervised" > gpurun_out/r03ad_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r03ad_tests.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert|FAILED" gpurun_out/r03ad_tests.log | head -30; exit $rc; fi
one() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
{
for wl in sup_r50 mono_r18 mono_r50; do
echo "$wl plain / force-overlap: $(one --workload $wl) $(one --workload $wl --force-overlap) $(one --workload $wl) $(one --workload $wl --force-overlap)"
done
} > gpurun_out/r03ad_tail.txt 2>&1
cat gpurun_out/r03ad_tail.txt
