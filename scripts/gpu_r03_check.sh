# Round-3 check: GPU tests, the default bench line, and rocprofv3 kernel stats of the same workload (eager) so that the bench line's
# roofline.avg_launch_us can be compared with the profiler's average for the same kernel family.   scripts/gpu_r03_check.sh TAG [notests]
TAG=${1:-r03a}
mkdir -p gpurun_out/prof
if [ "$2" != "notests" ]; then
  timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/${TAG}_gpu_tests.log 2>&1; rc=$?
  tail -3 gpurun_out/${TAG}_gpu_tests.log
  if [ $rc -ne 0 ]; then grep -E "FAILED|Error|error" gpurun_out/${TAG}_gpu_tests.log | head -20; exit $rc; fi
fi
timeout -k 10 300 python bench.py > gpurun_out/${TAG}_sup50_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -20 gpurun_out/${TAG}_bench.err; exit 4; }
cut -c1-1200 gpurun_out/${TAG}_sup50_bench.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o $TAG -- python3 bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --profile-steps 0 > gpurun_out/prof/${TAG}_eager.json 2> gpurun_out/prof/${TAG}_eager.err
echo "trace rc=$?"
f=$(ls gpurun_out/prof/*${TAG}_kernel_stats.csv | head -1); cp "$f" gpurun_out/${TAG}_sup50_kernel_stats.csv
python3 - "$TAG" <<'PY'
import csv, json, sys
tag = sys.argv[1]
line = json.loads(open(f'gpurun_out/{tag}_sup50_bench.json').read().strip().splitlines()[-1])
r = line['roofline']
tot = n = 0
for row in csv.DictReader(open(f'gpurun_out/{tag}_sup50_kernel_stats.csv')):
    if 'pgemm_kernel' in row['Name'] and 'Li64ELi64E' in row['Name']:
        tot += float(row['TotalDurationNs']); n += int(row['Calls'])
print(f"bench roofline: {r['kernel']} avg_launch_us {r['avg_launch_us']} median {r.get('median_launch_us')} frac {r['frac']} suspect {r.get('roofline_suspect')}")
print(f"rocprofv3     : pgemm<64,64> {n} dispatches, average {tot / n / 1e3:.2f} us  (ratio bench/rocprof {r['avg_launch_us'] / (tot / n / 1e3):.3f})")
PY
