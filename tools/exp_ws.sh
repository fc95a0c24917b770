#!/bin/bash
# Builds WITH the wave-specialised conv_h2w_kernel (tools/experiments/conv_h2w.inc; tag PLAIN = no ablation) and its ablations (never the product): tools/libexp_ws_<TAG>.so = the product objects with conv_h2.hip recompiled under
# -DLAD_WS_<TAG>: NOMFMA (M waves skip their MFMAs), NODMA (no weight ring traffic; with NOROWS the counted waits are vacuous), NOROWS (no row
# loads: constants), NOPUT (rows are loaded but not split / written).  Results are garbage; only the launch time means something.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
pkg=$root/laughter-detection-icsi_amd
defs=""
for t in "$@"; do defs="$defs -DLAD_WS_$t"; done
tag=$(echo "$@" | tr ' ' '_')
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops -DLAD_H2_WS_BUILD $defs -I $root/include -c $pkg/csrc/conv_h2.hip -o /tmp/conv_h2_ws_$tag.o
objs=$(ls $pkg/csrc/build/*.o | grep -v conv_h2.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/libexp_ws_$tag.so $objs /tmp/conv_h2_ws_$tag.o
echo built $root/tools/libexp_ws_$tag.so
