mkdir -p gpurun_out/final
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest_gpu.log 2>&1; rc=$?
tail -3 gpurun_out/final/pytest_gpu.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert|FAILED" gpurun_out/final/pytest_gpu.log | head -30; exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
