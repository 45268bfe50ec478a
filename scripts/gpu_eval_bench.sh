# evaluator microbenchmark + its kernel stats
mkdir -p gpurun_out/prof
python3 scripts/microbench_eval.py > gpurun_out/eval_bench.json 2> gpurun_out/eval_bench.err; echo "rc=$?"; cat gpurun_out/eval_bench.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o evalk -- python3 scripts/microbench_eval.py > /dev/null 2> gpurun_out/prof/evalk.err
echo "prof rc=$?"; f=$(ls gpurun_out/prof/*evalk_kernel_stats.csv | head -1); head -8 "$f" | cut -c1-220
