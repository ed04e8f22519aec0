#!/bin/bash
cd "$(dirname "$0")/.."
build() { touch realpdebench_amd/csrc/rpb_axg.hip; RPB_HIPCC_FLAGS="$*" python realpdebench_amd/build.py > /dev/null 2>&1 || echo "BUILD FAILED: $*"; }
for rep in 1 2; do for w in 8 12; do
  echo "== AXG_WAVES_XF=$w rep $rep"; build -DAXG_WAVES_XF=$w
  python tools/kbench.py axis 2>/dev/null | grep -E "lazy" | sed "s/^/   /"
  python bench.py --only-headline --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   step %.3f ms' % d['ms_per_step'])"
done; done
build
