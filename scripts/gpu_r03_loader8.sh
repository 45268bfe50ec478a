mkdir -p gpurun_out
for cfg in "4 2" "6 2" "8 2" "8 4" "3 1"; do set -- $cfg; echo "slots=$1 ahead=$2: $(SLOTS=$1 AHEAD=$2 timeout -k 10 40 python scripts/loader_gap_probe.py mono_r18 normal 2>/dev/null | tail -1)"; done > gpurun_out/r03x_gap_probe.txt
cat gpurun_out/r03x_gap_probe.txt
