# Round 3: the fused BatchNorm-backward reduction -- its tests first, then A/B bench lines inside one call (box-to-box variation is +-4 %).
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_nn.py -q -x -k "bn_backward_reduce or conv_batchnorm or bn_bwd" > gpurun_out/r03b_tests_bn.log 2>&1; rc=$?
tail -5 gpurun_out/r03b_tests_bn.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert" gpurun_out/r03b_tests_bn.log | head -30; exit $rc; fi
for i in 1 2; do
  for o in 1 0; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --profile-steps 0 --steps 40 --warmup 10 --opt bnbwd=$o > gpurun_out/r03b_bench_bnbwd${o}_$i.json 2>gpurun_out/r03b_err.log || { tail -5 gpurun_out/r03b_err.log; exit 5; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/r03b_bench_bnbwd${o}_$i.json").read().strip().splitlines()[-1])
print("bnbwd=$o run $i:", d["value"], "images/s", d["ms_per_step"], "ms/step")
PY
  done
done
timeout -k 10 900 python -m pytest tests -q -x -m gpu > gpurun_out/r03b_gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r03b_gpu_tests.log
if [ $rc -ne 0 ]; then grep -E "^FAILED|^ERROR|Error" gpurun_out/r03b_gpu_tests.log | head -20; fi
exit $rc
