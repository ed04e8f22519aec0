#!/bin/bash
# end-of-round evidence after the width-128 changes: full GPU test log, full bench line, fsi and Galerkin kernel tables -> gpurun_out/r05z_*
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -6 > gpurun_out/r05z_pytest_gpu.txt
timeout 900 python bench.py > gpurun_out/r05z_bench_line.json 2> gpurun_out/r05z_bench.err
timeout 300 python tools/fsi_probe.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp" > gpurun_out/r05z_fsi_kernel_table.txt
timeout 300 python tools/model_probe.py galerkin 2>&1 | grep -v "amdgpu.ids\|socket.cpp" > gpurun_out/r05z_galerkin_kernel_table.txt
