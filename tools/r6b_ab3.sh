#!/bin/bash
# A/B: CMX_ONE_EPILOGUE (eval launches: one copy of the epilogue), CMX_WG_ONE_EPILOGUE (wave pairs: masked epilogue only), on the GPU box
cd "$(dirname "$0")/.."
build() {
  files=$1; shift
  for f in $files; do touch realpdebench_amd/csrc/$f; done
  RPB_HIPCC_FLAGS="$*" python realpdebench_amd/build.py > /dev/null 2>&1 || echo "BUILD FAILED: $*"
}
step() { python bench.py --only-headline --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   step %.3f ms  loss_check %s' % (d['ms_per_step'], d.get('loss_check', {}).get('rel_err')))"; }
probe() {
  python tools/kbench.py cell_mix 2>/dev/null | grep -E "conv wgrad|lazy gelu|layer 0|eval" | sed "s/^/   /"
  step
  python tools/fwd_probe.py 32 2>/dev/null | grep -E "cell_mix|kernel time" | sed "s/^/   fwd /"
  RPB_ARITH=f16x2 python tools/fwd_probe.py 32 2>/dev/null | grep -E "cell_mix|kernel time" | sed "s/^/   fwd /"
}
for rep in 1 2; do
  echo "== ONE_EPILOGUE=0 WG_ONE_EPILOGUE=0, rep $rep"; build "rpb_cmx.hip" -DCMX_ONE_EPILOGUE=0 -DCMX_WG_ONE_EPILOGUE=0; probe
  echo "== ONE_EPILOGUE=1 WG_ONE_EPILOGUE=0, rep $rep"; build "rpb_cmx.hip" -DCMX_ONE_EPILOGUE=1 -DCMX_WG_ONE_EPILOGUE=0; probe
  echo "== ONE_EPILOGUE=1 WG_ONE_EPILOGUE=1, rep $rep"; build "rpb_cmx.hip" -DCMX_ONE_EPILOGUE=1 -DCMX_WG_ONE_EPILOGUE=1; probe
done
build "rpb_cmx.hip"
