#!/usr/bin/env python3
"""Round 5: which non-MFMA work hides behind which MFMA shape, placed where?  Generates tools/experiments/issue_overlap2.hip.

Round 4's issue_overlap.hip showed that a wave issuing back-to-back v_mfma_f32_16x16x32_f16 starves a SECOND wave of the same SIMD
completely (times add).  VERDICT r4 item 1 asks for what it did not test: (a) v_mfma_f32_32x32x16_f16, which holds the SIMD's vector
issue for 8 of its 32 cycles instead of 8 of 16 (MI355X_MICROARCH.md, row 'vector-instruction ISSUE cost'); (b) the fillers placed in
the SAME wave, between its MFMAs, which is what conv_h2 / wgrad_h2 could be rebuilt to do.

Every kernel body is ONE asm volatile block per loop iteration, so the instruction stream is exactly what is written here (checked
with llvm-objdump).  One workgroup per CU (96 KB of LDS each), 256 threads = one wave per SIMD, 512 = two.

configs (name -> what a wave executes per iteration; an iteration holds 512 matrix-pipe cycles: 32 x 16x16x32 or 16 x 32x32x16):
  same-wave:   s<shape>_v<n>   n v_fma_f32 after every MFMA          (n = 0..6 for 16x16x32, 0..12 for 32x32x16)
               s<shape>_d<n>   n ds_read_b128 after every MFMA (per 100: d50 = one every other MFMA)
               s<shape>_m      1 ds_read_b128 + k v_fma per MFMA = conv_h2's tap-loop mix plus a share of its staging work
  burst:       b<shape>_v<n>   the same instruction multiset, all MFMAs first, then all fillers (what the kernels do today)
  two waves per SIMD (w2) of the same program: in phase and (for burst) in anti-phase (waves 4-7 start with the fillers)
  cross-wave:  x<shape>_v / x<shape>_d : waves 0-3 only MFMAs, waves 4-7 only fillers (round 4's test, both shapes)

    python tools/experiments/gen_issue_overlap2.py && \
    hipcc -O3 --offload-arch=gfx950 tools/experiments/issue_overlap2.hip -o tools/experiments/issue_overlap2 && tools/experiments/issue_overlap2
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))

NV = 8   # rotating v_fma registers
ND = 4   # rotating ds_read destinations


def mfma(shape, i):
    if shape == 16:
        return f"v_mfma_f32_16x16x32_f16 %[c{i % 16}], %[a], %[b], %[c{i % 16}]"
    return f"v_mfma_f32_32x32x16_f16 %[c{i % 4}], %[a], %[b], %[c{i % 4}]"


class Ctr:
    def __init__(self):
        self.v = 0
        self.d = 0

    def valu(self):
        s = f"v_fma_f32 %[v{self.v % NV}], %[v{self.v % NV}], %[m], %[k]"
        self.v += 1
        return s

    def lds(self):
        s = f"ds_read_b128 %[t{self.d % ND}], %[addr] offset:{(self.d % 16) * 1024}"
        self.d += 1
        return s


def body_same(shape, nv100, nd100, burst=False, fillers_first=False):
    """n*100 fillers per 100 MFMAs, spread evenly (Bresenham)."""
    nm = 32 if shape == 16 else 16
    c = Ctr()
    ms, fs = [], []
    out = []
    av = ad = 0
    for i in range(nm):
        out.append(mfma(shape, i))
        ms.append(out[-1])
        av += nv100
        ad += nd100
        while ad >= 100:
            ad -= 100
            out.append(c.lds())
            fs.append(out[-1])
        while av >= 100:
            av -= 100
            out.append(c.valu())
            fs.append(out[-1])
    if burst:
        out = fs + ms if fillers_first else ms + fs
    if c.d:
        out.append("s_waitcnt lgkmcnt(0)")
    return out, nm, c.v, c.d


def asm_block(lines, shape, use_v, use_d):
    nacc = 16 if shape == 16 else 4
    outs = [f'[c{i}] "+v"(acc[{i}])' for i in range(nacc)]
    if use_v:
        outs += [f'[v{i}] "+v"(v[{i}])' for i in range(NV)]
    if use_d:
        outs += [f'[t{i}] "=&v"(t[{i}])' for i in range(ND)]
    ins = ['[a] "v"(a)', '[b] "v"(b)']
    if use_v:
        ins += ['[m] "v"(m)', '[k] "v"(k)']
    if use_d:
        ins += ['[addr] "v"(addr)']
    text = "\n".join(f'            "{l}\\n"' for l in lines)
    return f"        asm volatile(\n{text}\n            : {', '.join(outs)}\n            : {', '.join(ins)});\n"


KERNEL = """
extern "C" __global__ __launch_bounds__({nt}, 1) void {name}(float *out, int iters, unsigned long long *clk, int roles) {{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 24 * 1024 / 4; i += {nt}) reinterpret_cast<float *>(lds)[i] = hashf(i);
    __syncthreads();
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) {{
        a[j] = (_Float16)hashf(lane * 8 + j);
        b[j] = (_Float16)hashf(1000 + lane * 8 + j);
    }}
    {accdecl}
    float v[{NV}];
    for (int i = 0; i < {NV}; ++i) v[i] = hashf(77 + lane + i);
    f32x4 t[{ND}];
    for (int i = 0; i < {ND}; ++i) t[i] = f32x4{{0.f, 0.f, 0.f, 0.f}};
    const float m = 0.999f, k = 0.001f;
    const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) void *)lds + lane * 16;
    (void)v; (void)t; (void)m; (void)k; (void)addr;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
{loops}
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    {accsum}
    for (int i = 0; i < {NV}; ++i) s += v[i];
    for (int i = 0; i < {ND}; ++i) s += t[i][0];
    out[blockIdx.x * {nt} + threadIdx.x] = s;
    if (lane == 0) {{
        clk[(blockIdx.x * 8 + wave) * 2] = c1 - c0;
        clk[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0;
    }}
}}
"""


def kernel(name, nt, shape, loops):
    nacc = 16 if shape == 16 else 4
    if shape == 16:
        accdecl = f"f32x4 acc[{nacc}];\n    for (int i = 0; i < {nacc}; ++i) acc[i] = f32x4{{0.f, 0.f, 0.f, 0.f}};"
        accsum = f"for (int i = 0; i < {nacc}; ++i) s += acc[i][0] + acc[i][3];"
    else:
        accdecl = f"f32x16 acc[{nacc}];\n    for (int i = 0; i < {nacc}; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;"
        accsum = f"for (int i = 0; i < {nacc}; ++i) s += acc[i][0] + acc[i][15];"
    return KERNEL.format(name=name, nt=nt, accdecl=accdecl, accsum=accsum, loops=loops, NV=NV, ND=ND)


def loop(block, cond=None):
    s = f"    for (int it = 0; it < iters; ++it) {{\n{block}    }}\n"
    if cond:
        s = f"    if ({cond}) {{\n{s}    }}\n"
    return s


def main():
    cfgs = []   # (name, nt, shape, mfma/iter/wave, valu/iter, lds/iter, note)
    src = []
    for shape in (16, 32):
        per = 1 if shape == 16 else 2   # fillers are quoted per 16 matrix-pipe cycles, so n means the same work for both shapes
        # same-wave, evenly spread, one and two waves per SIMD
        for nv in (0, 1, 2, 3, 4, 6):
            for nt in (256, 512):
                lines, nm, cv, cd = body_same(shape, nv * 100 * per, 0)
                name = f"s{shape}_v{nv}_w{nt // 256}"
                src.append(kernel(name, nt, shape, loop(asm_block(lines, shape, cv > 0, False))))
                cfgs.append((name, nt, shape, nm, cv, cd, f"same wave, {nv} v_fma per 16 pipe cycles, spread"))
        for nd in (25, 33, 50, 100):
            for nt in (256, 512):
                lines, nm, cv, cd = body_same(shape, 0, nd * per)
                name = f"s{shape}_d{nd}_w{nt // 256}"
                src.append(kernel(name, nt, shape, loop(asm_block(lines, shape, False, True))))
                cfgs.append((name, nt, shape, nm, cv, cd, f"same wave, {nd / 100:.2f} ds_read_b128 per 16 pipe cycles, spread"))
        # conv_h2's tap loop (16 ds_read_b128 per 768 pipe cycles = 0.33 per 16) plus n v_fma per 16 pipe cycles of staging / epilogue work
        for nv in (1, 2, 3):
            for nt in (256, 512):
                lines, nm, cv, cd = body_same(shape, nv * 100 * per, 33 * per)
                name = f"s{shape}_m{nv}_w{nt // 256}"
                src.append(kernel(name, nt, shape, loop(asm_block(lines, shape, True, True))))
                cfgs.append((name, nt, shape, nm, cv, cd, f"same wave, 0.33 ds_read_b128 + {nv} v_fma per 16 pipe cycles, spread"))
        # burst: the same multiset, MFMAs first then fillers; two waves per SIMD in phase and in anti-phase
        for nv in (1, 2, 3):
            lines, nm, cv, cd = body_same(shape, nv * 100 * per, 0, burst=True)
            for nt in (256, 512):
                name = f"b{shape}_v{nv}_w{nt // 256}"
                src.append(kernel(name, nt, shape, loop(asm_block(lines, shape, True, False))))
                cfgs.append((name, nt, shape, nm, cv, cd, f"burst: all MFMAs then {nv} v_fma per 16 pipe cycles" + (", both waves in phase" if nt == 512 else "")))
            lines2, _, _, _ = body_same(shape, nv * 100 * per, 0, burst=True, fillers_first=True)
            name = f"b{shape}_v{nv}_anti"
            lp = loop(asm_block(lines, shape, True, False), "wave < 4") + loop(asm_block(lines2, shape, True, False), "wave >= 4")
            src.append(kernel(name, 512, shape, lp))
            cfgs.append((name, 512, shape, nm, cv, cd, f"burst, two waves per SIMD in ANTI-phase (waves 4-7 start with their {nv} v_fma per 16)"))
        # cross-wave (round 4's test): waves 0-3 MFMAs only, waves 4-7 fillers only; roles bit 0 = MFMA waves run, bit 1 = filler waves run
        nm = 32 if shape == 16 else 16
        ml = [mfma(shape, i) for i in range(nm)]
        c = Ctr()
        vl = [c.valu() for _ in range(64)]   # 64 v_fma per 512 pipe cycles = 2 per 16
        lp = loop(asm_block(ml, shape, False, False), "wave < 4 && (roles & 1)") + loop(asm_block(vl, shape, True, False), "wave >= 4 && (roles & 2)")
        src.append(kernel(f"x{shape}_v", 512, shape, lp))
        cfgs.append((f"x{shape}_v", 512, shape, nm, 64, 0, "CROSS-wave: waves 0-3 MFMAs, waves 4-7 64 v_fma per 512 pipe cycles"))
        c = Ctr()
        dl = [c.lds() for _ in range(16)] + ["s_waitcnt lgkmcnt(0)"]
        lp = loop(asm_block(ml, shape, False, False), "wave < 4 && (roles & 1)") + loop(asm_block(dl, shape, False, True), "wave >= 4 && (roles & 2)")
        src.append(kernel(f"x{shape}_d", 512, shape, lp))
        cfgs.append((f"x{shape}_d", 512, shape, nm, 0, 16, "CROSS-wave: waves 0-3 MFMAs, waves 4-7 16 ds_read_b128 per 512 pipe cycles"))

    table = ",\n".join(
        f'    {{"{n}", (const void *){n}, {nt}, {sh}, {nm}, {cv}, {cd}, "{note}"}}' for n, nt, sh, nm, cv, cd, note in cfgs)
    host = HOST.replace("@TABLE@", table)
    with open(os.path.join(HERE, "issue_overlap2.hip"), "w") as f:
        f.write(HEAD + "".join(src) + host)
    print(f"{len(cfgs)} kernels")


HEAD = """// GENERATED by tools/experiments/gen_issue_overlap2.py -- do not edit.  See that file for what is measured and why.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float hashf(int i) {   // full-range, both signs, in (-1, 1): the clock under load depends on the data
    unsigned x = (unsigned)i * 2654435761u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    return ((int)(x >> 8) - (1 << 23)) * (1.0f / (1 << 23));
}
"""

HOST = """
struct Cfg { const char *name; const void *fn; int nt, shape, nm, nv, nd; const char *note; };
static const Cfg cfgs[] = {
@TABLE@
};

int main(int argc, char **argv) {
    const char *only = argc > 1 ? argv[1] : nullptr;
    float *out;
    unsigned long long *clk, *hclk = new unsigned long long[256 * 16];
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&clk, 256 * 16 * 8);
    const int iters = 3000;   // 1.536 M matrix-pipe cycles per wave
    const int LDSB = 96 * 1024;   // one workgroup per CU
    printf("%-16s %4s %6s %9s %9s %8s %8s  %s\\n", "config", "w/S", "roles", "ms", "cyc/iter", "x floor", "MHz", "what");
    for (const Cfg &c : cfgs) {
        if (only && !strstr(c.name, only)) continue;
        hipFuncSetAttribute(c.fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
        const bool cross = c.name[0] == 'x';
        for (int roles = cross ? 1 : 3; roles <= 3; ++roles) {
            float best = 1e9f;
            std::vector<double> cyc, mhz;
            for (int rep = 0; rep < 4; ++rep) {
                hipMemset(clk, 0, 256 * 16 * 8);
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                int it = iters, ro = roles;
                void *args[] = {&out, &it, &clk, &ro};
                hipLaunchKernel(c.fn, dim3(256), dim3(c.nt), args, LDSB, 0);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                hipEventDestroy(e0); hipEventDestroy(e1);
                if (rep == 0) continue;   // warm-up
                best = std::min(best, ms);
                hipMemcpy(hclk, clk, 256 * 16 * 8, hipMemcpyDeviceToHost);
                for (int i = 0; i < 256 * 8; ++i)
                    if (hclk[2 * i + 1] > 1000) {   // (a wave that ran its loop)
                        cyc.push_back((double)hclk[2 * i] / iters);
                        mhz.push_back(100.0 * (double)hclk[2 * i] / (double)hclk[2 * i + 1]);
                    }
            }
            std::sort(cyc.begin(), cyc.end());
            std::sort(mhz.begin(), mhz.end());
            const double cm = cyc.empty() ? 0 : cyc[cyc.size() / 2], fm = mhz.empty() ? 0 : mhz[mhz.size() / 2];
            const double floor_c = 512.0 * (c.nt / 256);   // matrix-pipe cycles per iteration on one SIMD (both waves' MFMAs)
            printf("%-16s %4d %6d %9.3f %9.1f %8.3f %8.0f  %s\\n", c.name, c.nt / 256, roles, best, cm, cm / floor_c, fm, c.note);
            fflush(stdout);
        }
    }
    return 0;
}
"""

if __name__ == "__main__":
    main()
