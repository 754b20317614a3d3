#!/bin/bash
# Diagnostic builds of libgar_hip.so: one source rebuilt with extra -D flags, linked with the product objects.
#   tools/build_variant.sh <name> <source-stem> "<flags>"   ->  grasp-any-region_amd/gar_amd/variants/libgar_hip_<name>.so
# Use with GAR_HIP_LIB=<path> (gar_amd/hip.py). The product library never contains these switches.
set -e
name=$1; stem=$2; flags=$3
cd "$(dirname "$0")/../grasp-any-region_amd/csrc"
make -s
mkdir -p build/var_$name ../gar_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -Wno-unused-result $flags -c $stem.hip -o build/var_$name/$stem.o
objs=$(ls build/*.o | grep -v "build/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../gar_amd/variants/libgar_hip_$name.so $objs build/var_$name/$stem.o
echo built ../gar_amd/variants/libgar_hip_$name.so
