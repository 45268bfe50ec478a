# template for A/B runs of the experiment switches of DESIGN 6.1 inside ONE call (box-to-box variation is +-4 %)
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload $WL --steps $ST --warmup 3 --no-cpu-baseline --profile-steps 0 $EXTRA > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "failed $name"; tail -3 gpurun_out/ab.err; exit 1; }
  if grep -q "HSA_STATUS" gpurun_out/ab.err; then echo "fault $name"; exit 3; fi
  echo "$WL $name $(python -c "import json;d=json.load(open('gpurun_out/ab.json'));print(d['value'], d['ms_per_step'])")"
}
ST=8
WL=mono_packnet run g1 SDE_WGRAD_GROUP=1
WL=mono_packnet run g3_128 SDE_WGRAD_GROUP=3
WL=mono_packnet run g3_64 SDE_WGRAD_GROUP=3 SDE_WGRAD_GROUP_MAX_MB=64
WL=mono_packnet run g3_all SDE_WGRAD_GROUP=3 SDE_WGRAD_GROUP_MAX_MB=100000
WL=mono_packnet run g1 SDE_WGRAD_GROUP=1
ST=30
WL=sup_r50 run g3_128 SDE_WGRAD_GROUP=3
WL=sup_r50 run g3_64 SDE_WGRAD_GROUP=3 SDE_WGRAD_GROUP_MAX_MB=64
WL=sup_r50 run g3_32 SDE_WGRAD_GROUP=3 SDE_WGRAD_GROUP_MAX_MB=32
WL=sup_r50 run g3_all SDE_WGRAD_GROUP=3 SDE_WGRAD_GROUP_MAX_MB=100000
