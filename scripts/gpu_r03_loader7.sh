mkdir -p gpurun_out
{ timeout -k 10 200 python scripts/loader_host_phases.py mono_r18; } > gpurun_out/r03x_loader_host.txt 2>gpurun_out/r03x_loader_host.err
cat gpurun_out/r03x_loader_host.txt; tail -3 gpurun_out/r03x_loader_host.err
