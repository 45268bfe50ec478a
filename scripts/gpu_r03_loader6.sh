mkdir -p gpurun_out
mk() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 20 --marks "$@" 2>gpurun_out/marks.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], json.dumps(d['marks_us']))"; }
{
echo "mono_r18 loader:   $(mk --workload mono_r18 --with-loader)"
echo "sup_r50 loader:   $(mk --workload sup_r50 --with-loader)"
} > gpurun_out/r03x_loader_marks.txt 2>&1
cat gpurun_out/r03x_loader_marks.txt
