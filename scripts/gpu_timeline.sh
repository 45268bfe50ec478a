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
# phases of the LAST step: forward (up to the loss kernel: one queue, serial), backward (two queues: which one is the critical chain, how busy it
# is and where it stalls), tail (after the last data-gradient kernel: stem weight gradient, slab sums, optimizer)
last = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(last[0]['Start_Timestamp'])
S = lambda r: (int(r['Start_Timestamp']) - t0) / 1e3
E = lambda r: (int(r['End_Timestamp']) - t0) / 1e3
loss = next((i for i, r in enumerate(last) if ('silog_' in r['Kernel_Name'] and 'fwd' in r['Kernel_Name']) or 'photo_fwd' in r['Kernel_Name']), None)
if loss is not None:
    fw, bw = last[:loss], last[loss:]
    print(f'forward: {len(fw)} kernels, {E(fw[-1]):.1f} us span, {sum(E(r) - S(r) for r in fw):.1f} us busy')
    qs = collections.Counter(r['Queue_Id'] for r in bw)
    for qid, _ in qs.most_common():
        ch = [r for r in bw if r['Queue_Id'] == qid]
        busy = sum(E(r) - S(r) for r in ch)
        gaps = [(S(b_) - E(a_), fam(a_['Kernel_Name'])[:24], fam(b_['Kernel_Name'])[:24]) for a_, b_ in zip(ch, ch[1:]) if S(b_) - E(a_) > 3]
        ftot = collections.Counter()
        for r in ch: ftot[fam(r['Kernel_Name'])[:36]] += E(r) - S(r)
        print(f'backward queue {qid}: {len(ch)} kernels, {S(ch[0]):.1f}..{E(ch[-1]):.1f} us, busy {busy:.1f} us, {len(gaps)} stalls > 3 us adding up to {sum(g[0] for g in gaps):.1f} us')
        print('   largest: ' + ', '.join(f'{k} {v:.0f}' for k, v in ftot.most_common(8)))
        for g in sorted(gaps, reverse=True)[:5]:
            print(f'   stall {g[0]:6.1f} us  {g[1]} -> {g[2]}')
PY
