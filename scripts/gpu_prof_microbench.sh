# rocprofv3 kernel durations of the per-layer microbenchmark:  scripts/gpu_prof_microbench.sh TAG [MB_FILTER]
TAG=${1:-mb}
export MB_FILTER=${2:-}
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -o $TAG -- python3 scripts/microbench_conv.py > gpurun_out/prof/${TAG}.log 2> gpurun_out/prof/${TAG}.err
echo "rc=$?"
python3 - "$TAG" <<'PY'
import csv, glob, sys, re, collections
tag = sys.argv[1]
f = glob.glob(f'gpurun_out/prof/*{tag}_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# group launches by (kernel, grid) in order of first appearance
agg = collections.OrderedDict()
for r in rows:
    k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    k = re.sub(r'^void ', '', k)[:70]
    key = (k, r['Grid_Size_X'], r.get('Workgroup_Size_X', ''))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg.setdefault(key, []).append(d)
with open(f'gpurun_out/prof/{tag}_by_launch_shape.txt', 'w') as o:
    for (k, gx, wx), v in agg.items():
        v.sort()
        o.write(f'{k:70s} grid {gx:>8s} n {len(v):4d} median {v[len(v)//2]:8.1f} us min {v[0]:8.1f}\n')
print(open(f'gpurun_out/prof/{tag}_by_launch_shape.txt').read())
PY
