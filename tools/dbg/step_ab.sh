#!/bin/bash
# tools/dbg/step_ab.sh "base nopk2 ..." : FNO step + rollout of each library variant, alternating twice; prints ms_per_step and the rollout ms per forward
FNO="--no-pmc --no-bf16 --no-transolver --no-galerkin --no-unet --no-dpot --no-cpu-baseline --no-fno-native --no-scaling-proxy"
for rep in 1 2; do
for v in $1; do
  if [ $v = base ]; then unset RPB_LIB_PATH; else export RPB_LIB_PATH=$PWD/tools/dbg/librpb_$v.so; fi
  python bench.py $FNO 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l)
        r = d.get('rollout', {}) if isinstance(d.get('rollout'), dict) else {}
        print('$v', 'ms_per_step', d['ms_per_step'], 'rollout', {k: v for k, v in d.items() if 'rollout' in k and not isinstance(v, dict)}, {k: r[k] for k in list(r)[:6]})
"
done; done
