mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -v -m gpu > gpurun_out/gpu_tests.log 2>&1; rc=$?
grep -E "passed|failed|error" gpurun_out/gpu_tests.log | tail -5
if grep -q "HSA_STATUS_ERROR\|Aborted\|dumped core\|Fatal Python error" gpurun_out/gpu_tests.log; then echo "GPU fault in tests"; exit 3; fi
grep -E "^E  |FAILED|Error" gpurun_out/gpu_tests.log | head -30
timeout -k 10 600 python bench.py --workload mono_packnet --batch 4 --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 1 > gpurun_out/bench_packnet_b4.json 2> gpurun_out/bench_packnet_b4.err; echo "packnet b4 rc=$?"; tail -c 2500 gpurun_out/bench_packnet_b4.json; tail -3 gpurun_out/bench_packnet_b4.err
timeout -k 10 300 python bench.py --workload mono_r18 --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 2 > gpurun_out/bench_mono18.json 2> gpurun_out/bench_mono18.err; echo "mono18 rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/bench_mono18.json')); print(d['value'], d['ms_per_step'], d['roofline_photometric'])"
