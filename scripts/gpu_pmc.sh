# hardware counters per kernel (separate --pmc passes, kernel-trace only; see /opt/skills/guides/MI355X_MICROARCH.md)
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --profile-steps 0"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc -o sq -- python3 $ARGS > gpurun_out/pmc/sq.json 2> gpurun_out/pmc/sq.err; echo "sq rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d gpurun_out/pmc -o tcc -- python3 $ARGS > gpurun_out/pmc/tcc.json 2> gpurun_out/pmc/tcc.err; echo "tcc rc=$?"
ls -la gpurun_out/pmc | head -20
python3 - <<'PY'
import csv, glob, collections, re
def agg(path):
    d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(path)):
        k = re.sub(r'\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+', '', r['Kernel_Name'])[:48]
        d[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    return d
for f in glob.glob('gpurun_out/pmc/*counter_collection.csv'):
    print('==', f)
    d = agg(f)
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1].values()))[:16]:
        print(k, {a: f'{b:.3g}' for a, b in v.items()})
PY
