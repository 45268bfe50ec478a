mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0 $EXTRA > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "failed $name"; tail -3 gpurun_out/ab.err; exit 1; }
  if grep -q "HSA_STATUS" gpurun_out/ab.err; then echo "fault $name"; exit 3; fi
  echo "$WL $name $(python -c "import json;d=json.load(open('gpurun_out/ab.json'));print(d['value'], d['ms_per_step'])")"
}
WL=sup_r50 run warmup SDE_X=0
WL=sup_r50 run base SDE_X=0
WL=sup_r50 run blocks384 SDE_WGRAD_BLOCKS=384
WL=sup_r50 run blocks192 SDE_WGRAD_BLOCKS=192
WL=sup_r50 run defer0 SDE_DEFER_MAX_MB=0
WL=sup_r50 run defer8 SDE_DEFER_MAX_MB=8
WL=sup_r50 run bmg128 SDE_WGRAD_BMG=128
WL=sup_r50 run base SDE_X=0
WL=sup_r50 run splitk_t512 SDE_SPLITK_TILES=512 SDE_SPLITK_TARGET=1024
WL=sup_r50 run nohalo SDE_NO_HALO=1
