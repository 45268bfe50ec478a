# rocprofv3 evidence for one bench workload:  scripts/gpu_profile.sh TAG [WORKLOAD] [extra bench args]
#   1. kernel trace + stats of eager steps (every kernel its own dispatch)
#   2. HBM traffic counters, separate --pmc passes with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md "HBM")
#   3. SQ counters (wave cycles, waits, issue, VALU / LDS activity)
# Per-kernel sums land in gpurun_out/prof/*_by_kernel.csv; copy what is to be judged into profiles/.
TAG=${1:-r02}
WL=${2:-sup_r50}
shift; shift
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ARGS="bench.py --workload $WL --steps 4 --warmup 2 --no-graph --no-cpu-baseline --profile-steps 0 $@"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o $TAG -- python3 $ARGS > gpurun_out/prof/${TAG}_bench.json 2> gpurun_out/prof/${TAG}_bench.err
echo "trace rc=$?"; tail -c 300 gpurun_out/prof/${TAG}_bench.json
f=$(ls gpurun_out/prof/*${TAG}_kernel_stats.csv | head -1); head -30 "$f" | cut -c1-200
ARGS2="bench.py --workload $WL --steps 2 --warmup 1 --no-graph --no-cpu-baseline --profile-steps 0 $@"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof -o ${TAG}_fetch -- python3 $ARGS2 > gpurun_out/prof/fetch.json 2> gpurun_out/prof/fetch.err && echo "fetch ok" &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof -o ${TAG}_write -- python3 $ARGS2 > gpurun_out/prof/write.json 2> gpurun_out/prof/write.err && echo "write ok" &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d gpurun_out/prof -o ${TAG}_sq -- python3 $ARGS2 > gpurun_out/prof/sq.json 2> gpurun_out/prof/sq.err && echo "sq ok"
python3 - "$TAG" <<'PY'
import csv, glob, collections, re, sys
tag = sys.argv[1]
for f in sorted(glob.glob(f'gpurun_out/prof/*{tag}_*counter_collection.csv')):
    d = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])[:72]
        d[(k, r['Counter_Name'])][0] += float(r['Counter_Value']); d[(k, r['Counter_Name'])][1] += 1
    out = f.replace('counter_collection.csv', 'by_kernel.csv')
    with open(out, 'w') as o:
        o.write('kernel,counter,sum,launches,per_launch\n')
        for (k, c), (v, n) in sorted(d.items(), key=lambda kv: (kv[0][0], kv[0][1])):
            o.write(f'"{k}",{c},{v:.6g},{n},{v / n:.6g}\n')
    print('==', out)
PY
