mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_nn.py -q -x -k "residual_batchnorm or supervised or golden or graph_replay or batchnorm or bnbwd or two_phase" > gpurun_out/r03af_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r03af_tests.log
if [ $rc -ne 0 ] || grep -q "Memory access fault" gpurun_out/r03af_tests.log; then grep -E "Error|error|assert|FAILED|fault" gpurun_out/r03af_tests.log | head -30; exit 1; fi
one() { timeout -k 10 120 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
{
for wl in sup_r18 mono_r18 sup_r50; do
echo "$wl residual form / separate reduce pass: $(one --workload $wl) $(one --workload $wl --opt resbn=0) $(one --workload $wl) $(one --workload $wl --opt resbn=0)"
done
} > gpurun_out/r03af_resbn18.txt 2>&1
cat gpurun_out/r03af_resbn18.txt
