# encoder-only MFMA fraction from a rocprofv3 kernel trace:  scripts/gpu_encoder_fraction.sh TAG
TAG=${1:-encfrac}
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -o $TAG -- python3 scripts/encoder_mfma_fraction.py run > gpurun_out/prof/${TAG}.log 2> gpurun_out/prof/${TAG}.err
echo "rc=$?"; tail -2 gpurun_out/prof/${TAG}.log
python3 scripts/encoder_mfma_fraction.py report $(ls gpurun_out/prof/*${TAG}_kernel_trace.csv | head -1) | tee gpurun_out/prof/${TAG}_report.txt
