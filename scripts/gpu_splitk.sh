mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0 > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "failed $name"; tail -3 gpurun_out/ab.err; exit 1; }
  if grep -q "HSA_STATUS" gpurun_out/ab.err; then echo "fault $name"; exit 3; fi
  echo "$WL $name $(python -c "import json;d=json.load(open('gpurun_out/ab.json'));print(d['value'], d['ms_per_step'])")"
}
WL=sup_r50 run t384_768 SDE_X=0
WL=sup_r50 run t512_1024 SDE_SPLITK_TILES=512 SDE_SPLITK_TARGET=1024
WL=sup_r50 run t768_1536 SDE_SPLITK_TILES=768 SDE_SPLITK_TARGET=1536
WL=sup_r50 run t1024_2048 SDE_SPLITK_TILES=1024 SDE_SPLITK_TARGET=2048
WL=sup_r50 run t256_512 SDE_SPLITK_TILES=256 SDE_SPLITK_TARGET=512
WL=sup_r50 run t384_768 SDE_X=0
