# Round 3: device-side input pipeline tests, then A/B bench lines inside one call: resident inputs vs --with-loader, plain vs --force-overlap (the N > 1 step shape on one GPU)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_device_aug.py tests/test_data.py -q -x > gpurun_out/r03d_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r03d_tests.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert|FAILED" gpurun_out/r03d_tests.log | head -30; exit $rc; fi
run() { # tag, args...
  tag=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline --profile-steps 0 --steps 40 --warmup 10 "$@" > gpurun_out/r03d_bench_$tag.json 2> gpurun_out/r03d_err.log || { echo "$tag FAILED"; tail -5 gpurun_out/r03d_err.log; return 1; }
  python - "$tag" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r03d_bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:28s} {d['value']:9.1f} images/s {d['ms_per_step']:8.3f} ms/step", d.get("input_side", ""))
PY
}
run sup_r50 && run sup_r50_loader --with-loader && run sup_r50_overlap --force-overlap && run sup_r50_b && run sup_r50_loader_b --with-loader &&
run mono_r18 --workload mono_r18 && run mono_r18_loader --workload mono_r18 --with-loader && run mono_r18_overlap --workload mono_r18 --force-overlap &&
run mono_r50 --workload mono_r50 && run mono_r50_overlap --workload mono_r50 --force-overlap && run mono_r50_loader --workload mono_r50 --with-loader
