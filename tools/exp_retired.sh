#!/bin/bash
# A library with ONE product source replaced by its retired-variants snapshot (tools/experiments/retired/<name>_variants.hip: the file as
# it stood at the end of round 4, with the round-2 kernels and the lad_*_set_variant / LAD_*_VARIANT knobs that round 5 took out of the
# product):   tools/exp_retired.sh conv_b3 | wgrad_mfma   ->  LAD_HIP_LIB=tools/libexp_<name>_variants.so python tools/bench_conv.py ...
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1
pkg=$root/laughter-detection-icsi_amd
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops -I $root/include -I $pkg/csrc \
    -c $root/tools/experiments/retired/${name}_variants.hip -o /tmp/${name}_variants.o
objs=$(ls $pkg/csrc/build/*.o | grep -v "/$name.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/libexp_${name}_variants.so $objs /tmp/${name}_variants.o
echo built $root/tools/libexp_${name}_variants.so
