mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_photometric.py tests/test_gpu_models.py -q -x -k "photometric or monodepth2 or mono or posenet" > gpurun_out/r03q_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r03q_tests.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert|FAILED" gpurun_out/r03q_tests.log | head -20; exit $rc; fi
one() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for wl in mono_r18 mono_r50; do
echo "$wl multi-scale launch: $(one --workload $wl) $(one --workload $wl) ; per-scale launches: $(one --workload $wl --opt photo_multi=0) $(one --workload $wl --opt photo_multi=0)"
done
timeout -k 10 200 python bench.py --workload mono_r18 --no-cpu-baseline --steps 10 --warmup 5 > gpurun_out/r03q_bench_mono_r18.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r03q_bench_mono_r18.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); print(d.get('roofline_photometric'))"
