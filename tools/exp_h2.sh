#!/bin/bash
# Diagnostic builds of conv_h2.hip with pieces removed (never the product): tools/libexp_h2_<tag>.so = the product objects with
# conv_h2.o recompiled under -D<flag>.   tools/exp_h2.sh NOMFMA   ->  LAD_HIP_LIB=tools/libexp_h2_NOMFMA.so python tools/bench_h2.py --only h2v1
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tag=$1
pkg=$root/laughter-detection-icsi_amd
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops ${2:+-DLAD_H2_$tag=$2} $( [ -z "$2" ] && echo -DLAD_H2_$tag ) -I $root/include -c $pkg/csrc/conv_h2.hip -o /tmp/conv_h2_$tag.o
objs=$(ls $pkg/csrc/build/*.o | grep -v conv_h2.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/libexp_h2_$tag.so $objs /tmp/conv_h2_$tag.o
echo built $root/tools/libexp_h2_$tag.so
