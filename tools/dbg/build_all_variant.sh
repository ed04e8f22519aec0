#!/bin/bash
# tools/dbg/build_all_variant.sh NAME "FILES" [flags...]: recompile the listed sources (or ALL) under extra flags -> tools/dbg/librpb_NAME.so
set -e
NAME=$1; FILES=$2; shift 2
cd "$(dirname "$0")/../../realpdebench_amd/csrc"
mkdir -p /tmp/var_$NAME
[ "$FILES" = "ALL" ] && FILES=$(ls *.hip)
OBJS=""
for f in $(ls *.hip); do
  if echo " $FILES " | grep -q " $f "; then
    EXTRA=""; [ "$f" = "rpb_pjg.hip" ] && EXTRA="-fno-slp-vectorize"
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $EXTRA "$@" -c $f -o /tmp/var_$NAME/${f%.hip}.o ) &
    OBJS="$OBJS /tmp/var_$NAME/${f%.hip}.o"
    while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.5; done
  else
    OBJS="$OBJS build/${f%.hip}.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/dbg/librpb_$NAME.so $OBJS
echo built tools/dbg/librpb_$NAME.so
