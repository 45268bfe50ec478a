# rocprofv3 kernel trace of the bench step (eager launches so that every kernel is a separate dispatch), then HBM traffic counters
# (separate --pmc passes, kernel-trace only: FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md "HBM" section)
TAG=${1:-r1e}
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --profile-steps 0"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o $TAG -- python3 $ARGS > gpurun_out/prof/bench.json 2> gpurun_out/prof/bench.err
echo "trace rc=$?"; tail -c 300 gpurun_out/prof/bench.json
f=$(ls gpurun_out/prof/*${TAG}_kernel_stats.csv | head -1); head -30 "$f"
ARGS2="bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --profile-steps 0"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof -o ${TAG}_fetch -- python3 $ARGS2 > gpurun_out/prof/fetch.json 2> gpurun_out/prof/fetch.err; echo "fetch rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof -o ${TAG}_write -- python3 $ARGS2 > gpurun_out/prof/write.json 2> gpurun_out/prof/write.err; echo "write rc=$?"
python3 - <<'PY'
import csv, glob, collections, re
for f in sorted(glob.glob('gpurun_out/prof/*counter_collection.csv')):
    d = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])[:60]
        d[(k, r['Counter_Name'])][0] += float(r['Counter_Value']); d[(k, r['Counter_Name'])][1] += 1
    out = f.replace('counter_collection.csv', 'by_kernel.csv')
    with open(out, 'w') as o:
        o.write('kernel,counter,sum,launches,per_launch\n')
        for (k, c), (v, n) in sorted(d.items(), key=lambda kv: -kv[1][0]):
            o.write(f'"{k}",{c},{v:.6g},{n},{v / n:.6g}\n')
    print('==', out); print(open(out).read()[:1500])
PY
