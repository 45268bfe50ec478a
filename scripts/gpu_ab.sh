mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0 $EXTRA > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "failed $name"; tail -3 gpurun_out/ab.err; exit 1; }
  if grep -q "HSA_STATUS" gpurun_out/ab.err; then echo "fault $name"; exit 3; fi
  echo "$WL $name $(python -c "import json;d=json.load(open('gpurun_out/ab.json'));print(d['value'], d['ms_per_step'])")"
}
WL=sup_r50 run warmup SDE_X=0
WL=sup_r50 run gemm SDE_NO_C1=1
WL=sup_r50 run fwd SDE_C1_FWD=1
WL=sup_r50 run fwd_dgrad SDE_C1_FWD=1 SDE_C1_DGRAD=1
WL=sup_r50 run gemm SDE_NO_C1=1
WL=sup_r50 run fwd SDE_C1_FWD=1
WL=sup_r50 run fwd_dgrad SDE_C1_FWD=1 SDE_C1_DGRAD=1
WL=mono_r18 run gemm SDE_NO_C1=1
WL=mono_r18 run fwd SDE_C1_FWD=1
WL=mono_r18 run fwd_dgrad SDE_C1_FWD=1 SDE_C1_DGRAD=1
