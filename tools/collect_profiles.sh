#!/bin/bash
# Round evidence in one GPU call: full bench line, HIP-event kernel table, rocprofv3 kernel-trace summary of the same FNO command,
# PMC passes (SQ / request-size traffic / L2, then instruction mix / LDS / co-execution) over the kernel micro-benchmarks.
# Outputs: gpurun_out/<tag>_*.      tools/collect_profiles.sh r03
TAG=${1:-rXX}
ROOT=$(pwd)
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
FNO="--no-pmc --no-bf16 --no-transolver --no-galerkin --no-unet --no-dpot --no-cpu-baseline --no-fno-native"
python bench.py $FNO --profile-all > /dev/null 2> gpurun_out/${TAG}_hip_event_kernel_table.txt
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$TAG && rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG -- python $ROOT/bench.py $FNO > /tmp/prof_$TAG.log 2>&1)
DB=$(find /tmp/prof_$TAG -name "*_results.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/${TAG}_train_step_kernel_stats.txt
tools/pmc_run.sh $TAG python tools/kbench.py cell_mix proj bwd_row axis > gpurun_out/${TAG}_pmc.log 2>&1
tools/pmc_run2.sh ${TAG}b python tools/kbench.py proj bwd_row > gpurun_out/${TAG}_pmc2.log 2>&1
(echo "## eval forward, headline shape B=32"; python tools/fwd_probe.py 32; echo; echo "## eval forward, headline shape B=32, OPT-IN f16x2 arithmetic"; RPB_ARITH=f16x2 python tools/fwd_probe.py 32; echo; echo "## eval forward, combustion volume B=16 fp32 storage"; python tools/fwd_probe.py 16 comb;
 echo; echo "## eval forward, combustion volume B=16 bf16 storage"; python tools/fwd_probe.py 16 comb_bf16;
 echo; echo "## hipGraph replay vs eager launches"; for w in headline comb comb_bf16; do python tools/graph_probe.py $w; done) 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_rollout_kernel_table.txt
./tools/ubench/stream_pat > gpurun_out/${TAG}_ubench_stream_pat.txt 2>&1
./tools/ubench/mfma_peak > gpurun_out/${TAG}_ubench_mfma_peak.txt 2>&1
ls -la gpurun_out | tail -30
