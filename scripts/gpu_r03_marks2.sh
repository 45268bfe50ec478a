mkdir -p gpurun_out
mk() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 40 --warmup 10 --marks "$@" 2>gpurun_out/marks.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], json.dumps(d['marks_us']))"; }
{
echo "sup_r50 plain: $(mk --workload sup_r50)"
echo "sup_r50 force-overlap: $(mk --workload sup_r50 --force-overlap)"
} > gpurun_out/r03ac_marks.txt 2>&1
cat gpurun_out/r03ac_marks.txt; tail -2 gpurun_out/marks.err
