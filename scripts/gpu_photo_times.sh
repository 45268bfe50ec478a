# kernel durations of the photometric kernels in the MonoDepth2 R18 workload:  scripts/gpu_photo_times.sh TAG
TAG=${1:-ph}
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -o $TAG -- python3 bench.py --workload mono_r18 --steps 4 --warmup 2 --no-graph --no-cpu-baseline --profile-steps 0 > gpurun_out/prof/${TAG}.json 2> gpurun_out/prof/${TAG}.err
echo "rc=$?"; tail -c 200 gpurun_out/prof/${TAG}.json
python3 - "$TAG" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f'gpurun_out/prof/*{tag}_kernel_trace.csv')[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'photo_' in n or 'smooth_' in n:
        k = ('photo_fwd' if 'photo_fwd' in n else 'photo_bwd' if 'photo_bwd' in n else n[:40], r['Grid_Size_X'], r['Grid_Size_Y'])
        agg[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items()):
    v.sort(); print(f'{k[0]:40s} grid {k[1]:>6s} x {k[2]:>5s}  n {len(v):3d}  median {v[len(v)//2]:7.1f} us  min {v[0]:7.1f}')
PY
