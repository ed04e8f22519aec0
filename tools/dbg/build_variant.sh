#!/bin/bash
# tools/dbg/build_variant.sh NAME FILE.hip [extra hipcc flags]: a copy of the library with ONE source recompiled under extra flags
# (everything else from csrc/build) -> tools/dbg/librpb_NAME.so, to A/B with RPB_LIB_PATH
set -e
NAME=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../../realpdebench_amd/csrc"
EXTRA=""
[ "$SRC" = "rpb_pjg.hip" ] && EXTRA="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $EXTRA "$@" -c $SRC -o /tmp/var_$NAME.o
OBJS=$(ls build/*.o | grep -v "build/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/dbg/librpb_$NAME.so $OBJS /tmp/var_$NAME.o
echo built tools/dbg/librpb_$NAME.so
