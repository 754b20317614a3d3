#!/bin/bash
# Diagnostic builds of libgar_hip.so: one or more sources rebuilt with extra -D flags, linked with the product objects.
#   tools/build_variant.sh <name> <source-stem> "<flags>" [<source-stem> "<flags>" ...]
#        ->  grasp-any-region_amd/gar_amd/variants/libgar_hip_<name>.so
# A stem with a slash is a source OUTSIDE csrc/ (relative to the repo root, without .hip) that is added to the link:
#   tools/build_variant.sh lw tools/gemm_lw/gemm_lw "" gemm_pp "-DGAR_GEMM_LW_VARIANT"          (round-6 4-wave GEMM frame)
#   tools/build_variant.sh v4 tools/attn_v4/attention_v4 "" attention "-DGAR_ATTN_V4_VARIANT"   (round-5 4-wave attention)
# Use with GAR_HIP_LIB=<path> (gar_amd/hip.py). The product library never contains these switches.
set -e
name=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "$root/grasp-any-region_amd/csrc"
make -s
mkdir -p build/var_$name ../gar_amd/variants
objs=$(make -s print-objs | tr " " "\n")
extra=""
while [ $# -ge 2 ]; do
  stem=$1; flags=$2; shift 2
  case "$stem" in
    */*) src="$root/$stem.hip"; obj=build/var_$name/$(basename $stem).o ;;
    *)   src="$stem.hip"; obj=build/var_$name/$stem.o; objs=$(echo "$objs" | grep -v "build/$stem.o") ;;
  esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -Wno-unused-result -I. $flags -c $src -o $obj
  extra="$extra $obj"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../gar_amd/variants/libgar_hip_$name.so $objs $extra
echo built ../gar_amd/variants/libgar_hip_$name.so
