# the replayed step's real timeline from in-graph timestamp markers (no profiler) + plain A/B timings
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_photometric.py tests/test_gpu_models.py -q -x -k "photometric or monodepth2 or mono" > gpurun_out/r03v_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r03v_tests.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert|FAILED" gpurun_out/r03v_tests.log | head -20; exit $rc; fi
mk() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 40 --warmup 10 --marks "$@" 2>gpurun_out/marks.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], json.dumps(d['marks_us']))"; }
one() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
{
echo "mono_r18 per-scale: $(mk --workload mono_r18)"
echo "mono_r18 multi: $(mk --workload mono_r18 --opt photo_multi=1)"
echo "mono_r50 per-scale: $(mk --workload mono_r50)"
echo "mono_r50 multi: $(mk --workload mono_r50 --opt photo_multi=1)"
for wl in mono_r18 mono_r50; do
echo "$wl per-scale / multi (no markers): $(one --workload $wl) $(one --workload $wl --opt photo_multi=1) $(one --workload $wl) $(one --workload $wl --opt photo_multi=1)"
done
} > gpurun_out/r03v_marks.txt 2>&1
cat gpurun_out/r03v_marks.txt; tail -3 gpurun_out/marks.err
