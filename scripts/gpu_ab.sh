mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_photometric.py tests/test_gpu_models.py -q -m gpu 2>&1 | tail -4 > gpurun_out/kf_tests.log; cat gpurun_out/kf_tests.log
if grep -q "HSA_STATUS_ERROR\|Aborted\|dumped core\|Fatal Python error\|failed" gpurun_out/kf_tests.log; then echo "stop"; exit 3; fi
timeout -k 10 300 python bench.py --workload mono_r18 --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 2 > gpurun_out/ab.json 2> gpurun_out/ab.err; python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print(d['value'], d['ms_per_step'], d['roofline_photometric']['all'])"
