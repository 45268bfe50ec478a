mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_device_aug.py tests/test_data.py tests/test_trainer_batch.py -q -x > gpurun_out/r03e_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r03e_tests.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert|FAILED" gpurun_out/r03e_tests.log | head -30; exit $rc; fi
run() { tag=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline --profile-steps 0 --steps 40 --warmup 10 "$@" > gpurun_out/r03e_bench_$tag.json 2> gpurun_out/r03e_err.log || { echo "$tag FAILED"; tail -5 gpurun_out/r03e_err.log; return 1; }
  python - "$tag" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r03e_bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:28s} {d['value']:9.1f} images/s {d['ms_per_step']:8.3f} ms/step", d.get("input_side", ""))
PY
}
run sup_r50 && run sup_r50_loader --with-loader && run sup_r50_b && run sup_r50_loader_b --with-loader &&
run mono_r18 --workload mono_r18 && run mono_r18_loader --workload mono_r18 --with-loader && run mono_r50 --workload mono_r50 && run mono_r50_loader --workload mono_r50 --with-loader
timeout -k 10 300 python scripts/loader_host_profile.py sup_r50 short > gpurun_out/r03e_loader_prof.txt 2>&1; grep "ms/step" gpurun_out/r03e_loader_prof.txt
