#!/bin/bash
# Ablation builds of block_f16_strip_kernel from a PATCHED COPY of csrc/conv_f16.hip, with the phase stamps (-DLAD_STAMP):
#   NOREAD   taps 1..8 multiply the first tap's fragments again (no read-ahead)     NOHOUSE  no weight / output traffic inside the taps
#   NOMFMA   fragments are read, nothing is multiplied                              NOBAR    no barrier on taps 1..8
#   NOW / NOOUT / NOSTORE   halves of NOHOUSE: no weight traffic / no output traffic / output read from LDS but not stored
#   PLAIN    the stamped kernel as it is
# tools/exp_blk.sh NOREAD  ->  LAD_STAMP_LIB=tools/libexp_blk_NOREAD.so python tools/stamp_block.py     (results are garbage: times only)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
pkg=$root/laughter-detection-icsi_amd
tag=$1
python3 - "$pkg/csrc/conv_f16.hip" $pkg/csrc/_exp_blk_$tag.hip $tag <<'PY'
import sys
src, dst, tag = sys.argv[1:4]
s = open(src).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b, 1)
for t in tag.split('_'):
    if t == 'NOREAD':
        rep('if (tap + 1 < TAPS) read_k(tap + 1, ks);', '')
    elif t == 'NOHOUSE':
        rep('if (wave < 4) {   // in flight: tap t + 2', 'if (false) {   // in flight: tap t + 2')
        rep('                        if (storing) {\n', '                        if (false) {\n')
        rep('if (storing_prev && it > 0) {', 'if (false) {')
    elif t == 'NOW':      # no weight traffic inside the taps (the ring keeps its first three taps)
        rep('if (wave < 4) {   // in flight: tap t + 2', 'if (false) {   // in flight: tap t + 2')
    elif t == 'NOOUT':    # the previous output is neither read from LDS nor stored
        rep('                        if (storing) {\n', '                        if (false) {\n')
        rep('if (storing_prev && it > 0) {', 'if (false) {')
    elif t == 'NOSTORE':  # ... read from LDS, not stored
        rep('if (storing_prev && it > 0) {', 'if (false) {')
    elif t == 'NOMFMA':
        rep('acc[n][rt] = mfma32_f16(wf[ks][n], xf[rt][ks], acc[n][rt]);', 'asm volatile("" : "+v"(acc[n][rt]) : "v"(wf[ks][n]), "v"(xf[rt][ks]));')
    elif t == 'NOBAR':
        rep('else asm volatile("s_waitcnt lgkmcnt(11)\\n\\ts_barrier" ::: "memory");', 'else asm volatile("s_waitcnt lgkmcnt(11)" ::: "memory");')
    elif t != 'PLAIN':
        raise SystemExit('unknown tag ' + t)
open(dst, 'w').write(s)
PY
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DLAD_STAMP -Xclang -target-feature -Xclang -packed-fp32-ops -I $root/include -c $pkg/csrc/_exp_blk_$tag.hip -o /tmp/conv_f16_blk_$tag.o 2>&1 | grep -v "recognized feature\|warning\|^\s*[0-9]* |\|\^\|generated" || true
rm -f $pkg/csrc/_exp_blk_$tag.hip
objs=$(ls $pkg/csrc/build/*.o | grep -v "/conv_f16.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/libexp_blk_$tag.so $objs /tmp/conv_f16_blk_$tag.o
echo built $root/tools/libexp_blk_$tag.so
