#!/bin/bash
# build_dw_variant.sh NAME FLAGS... -> tools/ab/libeffdet_NAME.so: the tree's objects with dwconv.hip recompiled under FLAGS (A/B runs via EFFDET_HIP_LIB)
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
python -m efficientdet.pytorch_amd.build >/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c efficientdet/pytorch_amd/csrc/dwconv.hip -o /tmp/dwconv_$name.o
objs=$(ls efficientdet/pytorch_amd/csrc/build/*.o | grep -v dwconv.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libeffdet_$name.so $objs /tmp/dwconv_$name.o
echo built tools/ab/libeffdet_$name.so
