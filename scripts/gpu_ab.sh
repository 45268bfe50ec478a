mkdir -p gpurun_out
SDE_PACK_SPLIT=1 timeout -k 10 900 python -m pytest tests/test_gpu_models.py -q -m gpu -x 2>&1 | tail -2 > gpurun_out/kf_tests.log; cat gpurun_out/kf_tests.log
if grep -q "HSA_STATUS_ERROR\|Aborted\|dumped core\|Fatal Python error\|failed" gpurun_out/kf_tests.log; then echo "stop"; exit 3; fi
run() { name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0 $EXTRA > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "failed $name"; tail -3 gpurun_out/ab.err; exit 1; }
  if grep -q "HSA_STATUS" gpurun_out/ab.err; then echo "fault $name"; exit 3; fi
  echo "$WL $name $(python -c "import json;d=json.load(open('gpurun_out/ab.json'));print(d['value'], d['ms_per_step'])")"
}
WL=sup_r50 run warmup SDE_X=0
WL=sup_r50 run whole SDE_X=0
WL=sup_r50 run split SDE_PACK_SPLIT=1
WL=sup_r50 run whole SDE_X=0
WL=sup_r50 run split SDE_PACK_SPLIT=1
WL=mono_r18 run whole SDE_X=0
WL=mono_r18 run split SDE_PACK_SPLIT=1
