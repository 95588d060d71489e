#!/bin/bash
# Build libeffdet_hip.so of another git revision for same-box A/B runs (EFFDET_HIP_LIB=<path> selects it at run time):
#   tools/build_rev_lib.sh REV OUT.so
set -e
REV=$1; OUT=$(realpath -m $2); T=$(mktemp -d)
git archive $REV efficientdet/pytorch_amd/csrc include | tar -x -C $T
mkdir -p $T/o $(dirname $OUT)
for f in $T/efficientdet/pytorch_amd/csrc/*.hip; do
  x=""; [ $(basename $f) = postprocess.hip ] && x="-ffp-contract=off"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $x -c $f -o $T/o/$(basename $f .hip).o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $T/o/*.o
rm -rf $T; echo built $OUT
