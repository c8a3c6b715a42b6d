#!/bin/bash
# tools/exp/build_alt.sh <name> [make variables ...] — a SECOND libomg_hip.so beside the product one, built from the same sources with other make
# variables, in tools/exp/build/<name>/ (git-ignored artefacts; the directory travels with the gpurun snapshot, gpurun_out/ does not).
#   bash tools/exp/build_alt.sh gelu2 GELU2=1 DEV=1        ->  OMG_HIP_LIB=$PWD/tools/exp/build/gelu2/libomg_hip.so python tools/ksched_ab.py 25 3 k
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; shift
D=$ROOT/tools/exp/build/$NAME
mkdir -p $D
cp $ROOT/omg_amd/csrc/*.hip $ROOT/omg_amd/csrc/*.h $ROOT/omg_amd/csrc/*.inc $ROOT/omg_amd/csrc/Makefile $D/
# the Makefile's relative paths (../../include, ../../tools/exp) are written for omg_amd/csrc: same depth + 1 here
sed -i 's#\.\./\.\./include#../../../../include#g; s#\.\./\.\./tools/exp#../../../../tools/exp#g' $D/Makefile $D/common.h
make -C $D -j8 "$@" 2>&1 | grep -E "error|hipcc" | cut -c1-160
ls -la $D/libomg_hip.so
