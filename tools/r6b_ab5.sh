#!/bin/bash
cd "$(dirname "$0")/.."
build() { touch realpdebench_amd/csrc/rpb_cmx.hip; RPB_HIPCC_FLAGS="$*" python realpdebench_amd/build.py > /dev/null 2>&1 || echo "BUILD FAILED: $*"; }
for rep in 1 2; do
  for w in 8 12 16; do
    echo "== CMX_WAVES_DFT_SB=$w rep $rep"; build -DCMX_WAVES_DFT_SB=$w
    python tools/fwd_probe.py 16 comb_bf16 2>/dev/null | grep -E "cell_mix|kernel time" | sed "s/^/   /"
  done
done
build
