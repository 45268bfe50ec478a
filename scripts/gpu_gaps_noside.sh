# kernel-to-kernel gaps inside the replayed hipGraph: rocprofv3 kernel trace of the default (graph) bench, analysed per queue
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -o gaps -- python3 bench.py --no-side-stream --steps 6 --warmup 3 --no-cpu-baseline --profile-steps 0 > gpurun_out/prof/gaps_bench.json 2> gpurun_out/prof/gaps_bench.err
echo "rc=$?"; tail -c 200 gpurun_out/prof/gaps_bench.json
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof/*gaps_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
print(len(rows), 'dispatches; columns', list(rows[0].keys())[:14])
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last ~6 steps: take the final 40% of dispatches
tail = rows[int(len(rows) * 0.6):]
t0, t1 = int(tail[0]['Start_Timestamp']), int(tail[-1]['End_Timestamp'])
busy = 0; cur_end = t0; gaps = []
for r in tail:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s > cur_end: gaps.append(s - cur_end)
    busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
span = t1 - t0
print(f'span {span/1e6:.2f} ms, GPU busy (union of kernels) {busy/1e6:.2f} ms = {100*busy/span:.1f} %, idle gaps: n={len(gaps)} total {sum(gaps)/1e6:.2f} ms, median {sorted(gaps)[len(gaps)//2]/1e3:.2f} us')
dur = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in tail]
print(f'kernels {len(tail)}, mean duration {sum(dur)/len(dur)/1e3:.2f} us, < 5 us: {sum(d < 5000 for d in dur)}, < 10 us: {sum(d < 10000 for d in dur)}')
big = sorted(gaps)[-10:]; print('largest gaps us', [round(g/1e3,1) for g in big])
PY
