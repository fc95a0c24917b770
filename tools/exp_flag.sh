#!/bin/bash
# Diagnostic build: the product objects with ONE source recompiled under extra compiler flags -> tools/libexp_<tag>.so
#   tools/exp_flag.sh conv_h2 maxilp -mllvm -amdgpu-sched-strategy=max-ilp
#   LAD_HIP_LIB=tools/libexp_maxilp.so python tools/bench_h2.py --only h2v1
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
src=$1; tag=$2; shift 2
pkg=$root/laughter-detection-icsi_amd
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -I $root/include -c $pkg/csrc/$src.hip -o /tmp/${src}_$tag.o 2>&1 | grep -v "recognized feature" || true
objs=$(ls $pkg/csrc/build/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/libexp_$tag.so $objs /tmp/${src}_$tag.o
echo built $root/tools/libexp_$tag.so
