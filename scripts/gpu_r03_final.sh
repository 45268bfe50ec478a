# final evidence of round 3: bench lines of every workload (resident, with the loader, two-phase), rocprofv3 kernel statistics of the default command
mkdir -p gpurun_out/final gpurun_out/prof
F=gpurun_out/final
line() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'), (d.get('input_side') or {}).get('copy_stream'))" $1; }
timeout -k 10 400 python bench.py > $F/bench_sup_r50.json 2> $F/bench_sup_r50.err || { tail -5 $F/bench_sup_r50.err; exit 1; }
line $F/bench_sup_r50.json
for wl in mono_r18 mono_r50 sup_r18; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --steps 40 --warmup 10 > $F/bench_$wl.json 2> $F/bench_$wl.err || { tail -5 $F/bench_$wl.err; exit 1; }
  line $F/bench_$wl.json
done
timeout -k 10 300 python bench.py --workload mono_packnet --no-cpu-baseline --steps 10 --warmup 3 > $F/bench_mono_packnet.json 2> $F/bench_mono_packnet.err || { tail -5 $F/bench_mono_packnet.err; exit 1; }
line $F/bench_mono_packnet.json
for wl in sup_r50 mono_r18 mono_r50 sup_r18; do
  timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --profile-steps 0 --steps 100 --warmup 20 --with-loader > $F/bench_${wl}_loader.json 2> /dev/null && line $F/bench_${wl}_loader.json
  timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 10 --force-overlap > $F/bench_${wl}_overlap.json 2> /dev/null && line $F/bench_${wl}_overlap.json
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r03_final_sup50 -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 3 > $F/rocprof_sup50.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r03_final_mono18 -- python3 bench.py --workload mono_r18 --no-cpu-baseline --steps 5 --warmup 3 > $F/rocprof_mono18.log 2>&1
ls gpurun_out/prof | grep r03_final
