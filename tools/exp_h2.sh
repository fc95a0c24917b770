#!/bin/bash
# Builds of the RETIRED conv_h2 structures and diagnostics (never the product): tools/libexp_h2_<tag>.so = the product objects with
# conv_h2.o replaced by tools/experiments/retired/conv_h2_variants.hip (all the variants of rounds 4-5 behind lad_conv_h2_set_variant /
# LAD_H2_VARIANT: 384-row tiles, four ring slots, persistent, ring-less, eight waves) compiled under -DLAD_H2_<tag>[=value]
# (NOMFMA, NOLOAD, NOEPI, NOSPLIT, STAGGER n, FENCETEST; tag VARIANTS = no diagnostic macro).
#   tools/exp_h2.sh VARIANTS   ->  LAD_HIP_LIB=tools/libexp_h2_VARIANTS.so python tools/bench_h2.py
#   tools/exp_h2.sh NOMFMA     ->  LAD_HIP_LIB=tools/libexp_h2_NOMFMA.so python tools/power_probe.py
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tag=$1
pkg=$root/laughter-detection-icsi_amd
def=""
[ "$tag" != "VARIANTS" ] && def="-DLAD_H2_$tag${2:+=$2}"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops $def -I $root/include -I $pkg/csrc \
    -I $root/tools/experiments/retired -c $root/tools/experiments/retired/conv_h2_variants.hip -o /tmp/conv_h2_$tag.o
objs=$(ls $pkg/csrc/build/*.o | grep -v conv_h2.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/libexp_h2_$tag.so $objs /tmp/conv_h2_$tag.o
echo built $root/tools/libexp_h2_$tag.so
