# Timeline of the replayed hipGraph step:  scripts/gpu_timeline.sh TAG [bench args]
# rocprofv3 kernel trace of the default (graph) bench; per step: wall span, union-busy time, per-queue busy time, and per kernel family the
# EXCLUSIVE time (nothing else running on the GPU) against its total time -- what shortening that family can buy in wall time.
TAG=${1:-tl}
shift
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -o $TAG -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --profile-steps 0 "$@" > gpurun_out/prof/${TAG}_bench.json 2> gpurun_out/prof/${TAG}_bench.err
echo "rc=$?"; tail -c 300 gpurun_out/prof/${TAG}_bench.json
python3 - "$TAG" <<'PY' | tee gpurun_out/prof/${TAG}_timeline.txt
import csv, glob, collections, re, sys
tag = sys.argv[1]
f = glob.glob(f'gpurun_out/prof/*{tag}_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def fam(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    n = re.sub(r'^_ZN\d+_GLOBAL__N_1\d+|^_ZN7sdeconv\d+', '', n)
    return n[:52]
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a, b = idx[-3] + 1, idx[-1] + 1          # the last two replayed steps
step = rows[a:b]
ev = []
for i, r in enumerate(step):
    ev.append((int(r['Start_Timestamp']), 1, i)); ev.append((int(r['End_Timestamp']), 0, i))
ev.sort()
active = set(); last = ev[0][0]
excl = collections.Counter(); tot = collections.Counter(); union = 0; both = 0
for t, kind, i in ev:
    dt = t - last
    if active:
        union += dt
        if len(active) == 1: excl[fam(step[next(iter(active))]['Kernel_Name'])] += dt
        else: both += dt
    last = t
    if kind: active.add(i)
    else: active.discard(i)
for r in step: tot[fam(r['Kernel_Name'])] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
span = int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])
q = collections.Counter()
for r in step: q[r['Queue_Id']] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
n = 2
print(f'per step: span {span/n/1e6:.3f} ms, union busy {union/n/1e6:.3f} ms, >=2 kernels running {both/n/1e6:.3f} ms, kernel-time sum {sum(tot.values())/n/1e6:.3f} ms, kernels {len(step)//n}')
print('busy per queue (ms/step):', {k: round(v / n / 1e6, 3) for k, v in q.items()})
print(f'{"family":52s} {"total us":>9s} {"exclusive us":>12s}')
for k, v in sorted(tot.items(), key=lambda kv: -excl[kv[0]])[:40]:
    print(f'{k:52s} {v/n/1e3:9.1f} {excl[k]/n/1e3:12.1f}')
PY
