#!/bin/bash
# Ablation builds of wgrad_h2_kernel from a PATCHED COPY of csrc/wgrad_mfma.hip (the product source carries no diagnostic macro):
#   NOMFMA  the tap loop keeps its fragment reads but issues no MFMA        NOLOAD  no row / dy / x traffic (constants, no LDS-DMA)
#   NOBN    the BatchNorm-backward arithmetic of the DOBN variants is skipped (dy passes through)
#   OCC1    (round 6) ONE workgroup per CU: __launch_bounds__(THREADS, 1) -- the whole 512-entry register file for a wave -- and 256 groups
# tools/exp_wgrad.sh NOLOAD  ->  LAD_HIP_LIB=tools/libexp_wgrad_NOLOAD.so python tools/wgrad_probe.py      (results are garbage: times only)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
pkg=$root/laughter-detection-icsi_amd
tag=$1
python3 - "$pkg/csrc/wgrad_mfma.hip" /tmp/wgrad_$tag.hip $tag <<'PY'
import sys
src, dst, tag = sys.argv[1:4]
s = open(src).read()
a = s.index('template <bool INBN, int DOBN>\n__global__ __launch_bounds__(THREADS, 2) void wgrad_h2_kernel')
b = s.index('template <bool INBN, int DOBN>\nint launch_wgrad_h2(')
k = s[a:b]
if tag == 'NOMFMA':
    k = k.replace('acc[tap][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[pa], b[nt][pb], acc[tap][mt][nt], 0, 0, 0);',
                  'asm volatile("" ::"v"(a[pa]), "v"(b[nt][pb]));')
elif tag == 'NOLOAD':
    k = k.replace('''    auto fetch = [&](int tile) {
        const int q0 = tile * TK;''', '''    auto fetch = [&](int tile) {
        pin[0] = pin[1] = u32x4{0x3f800000u + (unsigned)tid, 0x3f000000u, 0x40000000u, 0x3f800000u};
        if (DOBN == 0) pdo[0] = pdo[1] = u32x4{0x3a800000u, 0x3a000000u + (unsigned)tid, 0x3b000000u, 0x3a800000u};
        return;
        const int q0 = tile * TK;''')
elif tag == 'NOBN':
    k = k.replace('''        if (DOBN == 2) {
            const float4 sc = cf(0), sh = cf(1);''', '''        if (true) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                pdo[u] = as_u4(d[u]);
                buf_store16(pdo[u], dc_r, row_off(q0 + prow + RPP * u));
            }
            return;
        }
        if (DOBN == 2) {
            const float4 sc = cf(0), sh = cf(1);''')
elif tag == 'OCC1':
    k = k.replace('__global__ __launch_bounds__(THREADS, 2) void wgrad_h2_kernel', '__global__ __launch_bounds__(THREADS, 1) void wgrad_h2_kernel')
    tail = s[b:]
    c = tail.index('int launch_wgrad_h2(')
    d = tail.index('template <int CH, bool INBN>\nint launch_wgrad_b3(')
    tail = tail[:c] + tail[c:d].replace('const int groups = groups_for(n_tiles);', 'const int groups = std::min(groups_for(n_tiles), 256);') + tail[d:]
    s = s[:b] + tail
else:
    raise SystemExit('unknown tag')
assert k != s[a:b], 'patch did not apply'
open(dst, 'w').write(s[:a] + k + s[b:])
PY
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops -I $root/include -I $pkg/csrc -c /tmp/wgrad_$tag.hip -o /tmp/wgrad_$tag.o
objs=$(ls $pkg/csrc/build/*.o | grep -v wgrad_mfma.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/libexp_wgrad_$tag.so $objs /tmp/wgrad_$tag.o
echo built $root/tools/libexp_wgrad_$tag.so
