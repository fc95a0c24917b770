"""GPU tests of the "f16 x 2" split-operand kernels (csrc/conv_h2.hip, csrc/wgrad_h2.hip): the 64 -> 64 / 32 -> 32 3x3 stride-1
convolutions of models.py:86-96,110-115 on two f16 planes per operand.  Same bars as the bf16 x 3 kernels they replace on the
training path (tests/test_resnet_gpu.py): 2e-4 of max against torch, 5e-6 of max against the exact-f32 MFMA kernel; plus the
error against float64 next to the exact-f32 kernel's own."""
import struct

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_resnet_gpu import _lib, act_rows, borders_are_zero, from_pnhwc, to_pnhwc

pytestmark = pytest.mark.gpu


def pack_h2(h, lib, w, mode, C):
    """One packed image through lad_conv_h2_pack_weights_multi (a one-record table)."""
    wt = torch.zeros(int(lib.lad_conv_h2_packed_weight_bytes(C)), device="cuda", dtype=torch.uint8)
    rec = struct.pack("<QQii", w.data_ptr(), wt.data_ptr(), mode, 0)
    table = torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda()
    h.check(lib.lad_conv_h2_pack_weights_multi(h.ptr(table), 1, C, h.stream_handle()), "lad_conv_h2_pack_weights_multi")
    torch.cuda.synchronize()
    return wt


def unpack_h2(wt, C):
    """The packed image back as float64 weights [tap][k][n] (sum of the two planes, scale undone) and the stage exponents."""
    nstage, nct = C // 32, C // 16
    img = 9 * nstage * 2 * nct * 1024
    planes = wt[:img].view(torch.float16).view(9, nstage, 2, nct, 4, 16, 8).double().cpu()
    kexp = wt[img:img + 4 * nstage].view(torch.int32).cpu()
    w = planes.sum(2)                                              # [tap][stage][ct][kq][n][e]
    w = w * (2.0 ** (-kexp.double())).view(1, nstage, 1, 1, 1, 1)
    w = w.permute(0, 1, 3, 5, 2, 4).reshape(9, C, C)                # k = stage*32 + kq*8 + e ; n = ct*16 + n
    return w, kexp


@pytest.mark.parametrize("C", [64, 32])
def test_packed_weights_hold_the_weights_to_two_f16_planes(C):
    """h1 + h2 reproduces w 2^k to 2^-22 relative (two round-to-nearest f16 planes) with the stage's largest magnitude in
    [2^14, 2^15); mode 1 is the transposed, tap-flipped image; an all-zero stage is packed without overflow."""
    h = _lib()
    lib = h.lib()
    g = torch.Generator().manual_seed(3 + C)
    w = torch.randn(C, C, 3, 3, generator=g) * torch.exp(torch.randn(C, C, 1, 1, generator=g))
    w[:, 40 % C] = 0.0
    if C == 64:
        w[:, 32:] *= 1e-6          # the second K stage of the forward image lives 20 binades below the first
    wg = w.cuda()
    for mode in (0, 1):
        got, kexp = unpack_h2(pack_h2(h, lib, wg, mode, C), C)
        ref = w.double().permute(2, 3, 1, 0).reshape(9, C, C) if mode == 0 else w.double().flip(2, 3).permute(2, 3, 0, 1).reshape(9, C, C)
        err = (got - ref).abs()
        for s in range(C // 32):
            blk = ref[:, s * 32:(s + 1) * 32]
            amax = float(blk.abs().max())
            assert 2.0 ** 14 <= amax * 2.0 ** int(kexp[s]) < 2.0 ** 15
            # relative 2^-22 for everything within 2^17 of the stage's maximum, absolute 2^-25 (scaled) below
            bound = torch.maximum(blk.abs() * 2.0 ** -22, torch.full_like(blk, 2.0 ** (-25 - int(kexp[s]))))
            assert bool((err[:, s * 32:(s + 1) * 32] <= bound).all())
    z = torch.zeros(C, C, 3, 3, device="cuda")
    got, kexp = unpack_h2(pack_h2(h, lib, z, 0, C), C)
    assert float(got.abs().max()) == 0.0 and bool((kexp == 100).all())


@pytest.mark.parametrize("C,B,H,W", [(64, 3, 13, 6), (64, 2, 25, 11), (64, 29, 100, 44), (64, 5, 7, 46), (64, 1, 1, 1),
                                     (32, 3, 13, 6), (32, 40, 50, 22), (32, 5, 7, 46), (32, 1, 1, 1),
                                     # >= 1024 tiles of 256 rows: the launcher takes the 256-row kernel (two workgroups per CU);
                                     # everything above runs on the 128-row kernel (three per CU)
                                     (64, 64, 100, 44), (32, 230, 50, 22)])
def test_conv_h2_matches_the_f32_convolution(C, B, H, W):
    """Forward (+ bias + addend + BatchNorm partials + zero borders) and data gradient against torch (2e-4 of max) and the
    exact-f32 MFMA kernel (5e-6 of max) -- the bars of test_conv_b3_matches_the_f32_convolution, unchanged -- and the error
    against float64 is no more than 1.5x the exact-f32 kernel's own (both are dominated by the f32 accumulation)."""
    h = _lib()
    lib = h.lib()
    if True:
        g = torch.Generator().manual_seed(B * 1000 + H + C)
        x = torch.randn(B, C, H, W, generator=g)
        w = torch.randn(C, C, 3, 3, generator=g) * 0.1
        bias = torch.randn(C, generator=g)
        add = torch.randn(B, C, H, W, generator=g)
        st = h.stream_handle()
        wg, bg = w.cuda(), bias.cuda()
        rows = act_rows(B, H, W)
        n_tiles = int(lib.lad_conv_num_tiles(B, H, W))
        xin, addg = to_pnhwc(x), to_pnhwc(add)
        for mode in (0, 1):
            if mode == 0:
                ref64 = F.conv2d(x.double(), w.double(), bias.double(), padding=1) + add.double()
            else:
                ref64 = F.conv_transpose2d(x.double(), w.double(), padding=1)
            ref = ref64.float()
            wt = pack_h2(h, lib, wg, mode, C)
            wt_f = torch.zeros(int(lib.lad_conv_packed_weight_floats(C, C, 9, mode)), device="cuda")
            h.check(lib.lad_conv_pack_weights(h.ptr(wg), C, C, 9, mode, h.ptr(wt_f), st))
            out = torch.full((rows * C,), 9.0, device="cuda")
            out32 = torch.full((rows * C,), 9.0, device="cuda")
            part = torch.zeros(n_tiles * 2 * C, device="cuda")
            part32 = torch.zeros(n_tiles * 2 * C, device="cuda")
            b, a = (h.ptr(bg), h.ptr(addg)) if mode == 0 else (None, None)
            h.check(lib.lad_conv_h2(h.ptr(xin), None, h.ptr(wt), b, a, None, h.ptr(out), h.ptr(part), None, None, None, B, H, W, C, st),
                    "lad_conv_h2")
            h.check(lib.lad_conv_fwd(h.ptr(xin), h.ptr(wt_f), b, a, h.ptr(out32), h.ptr(part32), B, H, W, C, C, 9, st))
            got = from_pnhwc(out, B, C, H, W)
            scale = ref.abs().max().item()
            assert torch.allclose(got, ref, atol=2e-4 * scale), (mode, (got - ref).abs().max())
            assert (out - out32).abs().max().item() <= 5e-6 * scale, (mode, (out - out32).abs().max().item() / scale)
            assert borders_are_zero(out, B, C, H, W)
            e_h2 = float((got.double() - ref64).pow(2).mean().sqrt())
            e_32 = float((from_pnhwc(out32, B, C, H, W).double() - ref64).pow(2).mean().sqrt())
            assert e_h2 <= 1.5 * e_32 + 1e-12 * scale, (mode, e_h2, e_32)
            ps, ps32 = part.view(n_tiles, 2, C).double().sum(0), part32.view(n_tiles, 2, C).double().sum(0)
            assert torch.allclose(ps, ps32, rtol=1e-4, atol=1e-4 * float(ps32.abs().max()))
            # the launch is reproducible bit for bit
            out_b = torch.full((rows * C,), 3.0, device="cuda")
            part_b = torch.zeros(n_tiles * 2 * C, device="cuda")
            h.check(lib.lad_conv_h2(h.ptr(xin), None, h.ptr(wt), b, a, None, h.ptr(out_b), h.ptr(part_b), None, None, None, B, H, W, C, st))
            assert torch.equal(out, out_b) and torch.equal(part, part_b)


def test_conv_h2_block_scaling_over_a_wide_dynamic_range():
    """Tiles, channel stages and images of very different magnitude (1e-30 ... 1e+30, all-zero stages, one huge outlier): the
    per-tile power-of-two scales keep every region at fp32-level accuracy RELATIVE TO ITS OWN OUTPUT, where one scale per
    tensor could not; nothing overflows."""
    h = _lib()
    lib = h.lib()
    st = h.stream_handle()
    C, B, H, W = 64, 12, 100, 44
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, C, H, W, generator=g)
    mags = [1.0, 1e-30, 1e30, 1e-10, 1e10, 3e-5, 1.0, 1.0, 1.0, 1e-20, 1e20, 1.0]
    for i, s in enumerate(mags):
        x[i] *= s
    x[6, :32] = 0.0                 # an all-zero first stage
    x[7, 32:] *= 1e-12              # second stage 40 binades below the first
    x[8, 32:] *= 1e12               # ... and above
    x[11, 5, 50, 20] = 3e4          # one outlier 2^15 above its tile
    w = torch.randn(C, C, 3, 3, generator=g) * 0.1
    wg = w.cuda()
    wt = pack_h2(h, lib, wg, 0, C)
    rows = act_rows(B, H, W)
    out = torch.zeros(rows * C, device="cuda")
    h.check(lib.lad_conv_h2(h.ptr(to_pnhwc(x)), None, h.ptr(wt), None, None, None, h.ptr(out), None, None, None, None, B, H, W, C, st))
    got = from_pnhwc(out, B, C, H, W).double()
    assert bool(torch.isfinite(got).all())
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    for i in range(B):
        # interior of the image (its first / last rows share tiles with the neighbouring images of other magnitudes)
        sl = slice(12, 88)
        err = float((got[i, :, sl] - ref[i, :, sl]).abs().max())
        scale = float(ref[i, :, sl].abs().max())
        assert err <= 2e-6 * scale, (i, mags[i], err / scale)


@pytest.mark.parametrize("C,B,H,W", [(64, 3, 13, 6), (64, 29, 100, 44), (64, 5, 7, 46), (64, 1, 1, 1), (32, 9, 50, 22)])
def test_conv_h2_with_the_batchnorm_relu_applied_while_staging(C, B, H, W):
    """in_coef: the launch reads the previous convolution's raw output and applies BatchNorm + ReLU + the zero border while
    staging -- bit-identical to lad_bn_act followed by the plain launch (lad_conv_b3c_fwd_f32_bnrelu's contract)."""
    h = _lib()
    lib = h.lib()
    st = h.stream_handle()
    g = torch.Generator().manual_seed(B * 7 + H)
    rows, cnt = act_rows(B, H, W), B * H * W
    n_tiles = int(lib.lad_conv_num_tiles(B, H, W))
    c1 = to_pnhwc(torch.randn(B, C, H, W, generator=g) * 2 + 1)
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.1).cuda()
    bias = torch.randn(C, generator=g).cuda()
    gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.5 + 0.3).cuda()
    xn = from_pnhwc(c1, B, C, H, W).double()
    stat = torch.stack([xn.sum((0, 2, 3)), (xn ** 2).sum((0, 2, 3))]).float().reshape(-1).cuda()
    coef = torch.zeros(6 * C, device="cuda")
    h.check(lib.lad_bn_finalize(h.ptr(stat), 1, C, cnt, h.ptr(gam), h.ptr(bet), None, None, 0.1, h.ptr(coef), st))
    a1 = torch.zeros(rows * C, device="cuda")
    h.check(lib.lad_bn_act(h.ptr(c1), h.ptr(coef), None, None, h.ptr(a1), B, H, W, C, 1, st))
    wt = pack_h2(h, lib, w, 0, C)
    o1, o2 = torch.full((rows * C,), 5.0, device="cuda"), torch.full((rows * C,), 5.0, device="cuda")
    p1, p2 = torch.zeros(n_tiles * 2 * C, device="cuda"), torch.zeros(n_tiles * 2 * C, device="cuda")
    h.check(lib.lad_conv_h2(h.ptr(a1), None, h.ptr(wt), h.ptr(bias), None, None, h.ptr(o1), h.ptr(p1), None, None, None, B, H, W, C, st))
    h.check(lib.lad_conv_h2(h.ptr(c1), h.ptr(coef), h.ptr(wt), h.ptr(bias), None, None, h.ptr(o2), h.ptr(p2), None, None, None, B, H, W, C, st))
    assert float(o1.abs().max()) > 0
    assert torch.equal(o1, o2) and torch.equal(p1, p2)


@pytest.mark.parametrize("B,H,W", [(3, 13, 6), (29, 100, 44), (1, 1, 1)])
def test_conv_h2_data_gradient_with_gate_and_batchnorm_sums(B, H, W):
    """The data-gradient options of lad_conv_h2 against the bf16 x 3 entry points they replace: gated addend (sign bits),
    in place, and the BatchNorm-backward sums in the epilogue (mask from bits or recomputed): the epilogue is the shared one, so
    with equal convolution results the outputs would be bit-equal; here the convolution differs in rounding only -- outputs
    within 5e-6 of max, sums within 2e-5 of their largest."""
    h = _lib()
    lib = h.lib()
    st = h.stream_handle()
    C = 64
    g = torch.Generator().manual_seed(B * 13 + W)
    rows, cnt = act_rows(B, H, W), B * H * W
    n_tiles = int(lib.lad_conv_num_tiles(B, H, W))
    dout = to_pnhwc(torch.randn(B, C, H, W, generator=g))
    dy = to_pnhwc(torch.randn(B, C, H, W, generator=g))
    x = to_pnhwc(torch.randn(B, C, H, W, generator=g) * 2 + 1)
    res = to_pnhwc(torch.randn(B, C, H, W, generator=g))
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.1).cuda()
    gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()
    xn = from_pnhwc(x, B, C, H, W).double()
    stat = torch.stack([xn.sum((0, 2, 3)), (xn ** 2).sum((0, 2, 3))]).float().reshape(-1).cuda()
    coef = torch.zeros(6 * C, device="cuda")
    h.check(lib.lad_bn_finalize(h.ptr(stat), 1, C, cnt, h.ptr(gam), h.ptr(bet), None, None, 0.1, h.ptr(coef), st))
    y = torch.zeros(rows * C, device="cuda")
    ybits = torch.zeros(rows, device="cuda", dtype=torch.int64)
    h.check(lib.lad_bn_act_bits(h.ptr(x), h.ptr(coef), h.ptr(res), None, h.ptr(y), h.ptr(ybits), B, H, W, C, st))
    abits = torch.randint(-2 ** 62, 2 ** 62, (rows,), generator=g).cuda()
    wt3 = torch.zeros(int(lib.lad_conv_b3_packed_weight_bytes()), device="cuda", dtype=torch.uint8)
    h.check(lib.lad_conv_b3_pack_weights(h.ptr(w), 1, h.ptr(wt3), st))
    wt = pack_h2(h, lib, w, 1, C)
    for use_bits, gated in ((False, False), (True, True), (True, False)):
        o1, o2, o3 = (torch.zeros(rows * C, device="cuda") for _ in range(3))
        part1, part2 = torch.zeros(n_tiles * 2 * C, device="cuda"), torch.zeros(n_tiles * 2 * C, device="cuda")
        add, ab = (h.ptr(dy), h.ptr(abits)) if gated else (None, None)
        bb = h.ptr(ybits) if use_bits else None
        h.check(lib.lad_conv_b3_dgrad_bnstat(h.ptr(dout), h.ptr(wt3), add, ab, h.ptr(o1), h.ptr(part1), h.ptr(x), bb, h.ptr(coef), B, H, W, st))
        h.check(lib.lad_conv_h2(h.ptr(dout), None, h.ptr(wt), None, add, ab, h.ptr(o2), h.ptr(part2), h.ptr(x), bb, h.ptr(coef), B, H, W, C, st))
        h.check(lib.lad_conv_h2(h.ptr(dout), None, h.ptr(wt), None, add, ab, h.ptr(o3), None, None, None, None, B, H, W, C, st))
        scale = float(o1.abs().max())
        assert float((o1 - o2).abs().max()) <= 5e-6 * scale
        assert torch.equal(o2, o3)                       # the sums ride along: the output does not change
        s1, s2 = part1.view(n_tiles, 2, C).double().sum(0), part2.view(n_tiles, 2, C).double().sum(0)
        assert float((s1 - s2).abs().max()) <= 2e-5 * float(s1.abs().max()) + 1e-30
        if gated:
            inplace = dy.clone()
            h.check(lib.lad_conv_h2(h.ptr(dout), None, h.ptr(wt), None, h.ptr(inplace), ab, h.ptr(inplace), None, None, None, None, B, H, W, C, st))
            assert torch.equal(inplace, o3)
    # argument checks
    o = torch.zeros(rows * C, device="cuda")
    assert lib.lad_conv_h2(h.ptr(dout), None, h.ptr(wt), None, None, h.ptr(abits), h.ptr(o), None, None, None, None, B, H, W, C, st) != 0
    assert lib.lad_conv_h2(h.ptr(dout), None, h.ptr(wt), None, None, None, h.ptr(dout), None, None, None, None, B, H, W, C, st) != 0
    assert lib.lad_conv_h2(h.ptr(dout), None, h.ptr(wt), None, None, None, h.ptr(o), None, None, None, None, B, H, W, 16, st) != 0
    assert lib.lad_conv_h2(h.ptr(dout), None, h.ptr(wt), None, None, None, h.ptr(o), None, None, None, None, B, H, 47, C, st) != 0


@pytest.mark.parametrize("B,H,W", [(3, 13, 6), (2, 25, 11), (29, 100, 44), (5, 7, 46), (1, 1, 1), (40, 50, 22)])
def test_wgrad_h2_matches_the_f32_weight_gradient(B, H, W):
    """64 x 64 x 9 weight + bias gradient on two f16 planes per operand against torch autograd in float64 (2e-4 of max), the
    exact-f32 MFMA kernel (1e-5 of max: the bars of test_wgrad_b3_matches_the_f32_weight_gradient) and float64 (error no more
    than 1.5x the exact-f32 kernel's)."""
    h = _lib()
    lib = h.lib()
    C = 64
    g = torch.Generator().manual_seed(B * 77 + W)
    x = torch.randn(B, C, H, W, generator=g)
    dout = torch.randn(B, C, H, W, generator=g) * torch.exp(torch.randn(1, C, 1, 1, generator=g))   # per-channel scales
    st = h.stream_handle()
    xin, doutg = to_pnhwc(x), to_pnhwc(dout)
    ws = torch.zeros(int(lib.lad_conv_wgrad_workspace_floats(C, C, 9)), device="cuda")
    dw, db = torch.zeros(C, C, 3, 3, device="cuda"), torch.zeros(C, device="cuda")
    dw32, db32 = torch.zeros(C, C, 3, 3, device="cuda"), torch.zeros(C, device="cuda")
    h.check(lib.lad_conv_wgrad_h2(h.ptr(xin), None, h.ptr(doutg), h.ptr(ws), h.ptr(dw), h.ptr(db), B, H, W, C, st), "lad_conv_wgrad_h2")
    h.check(lib.lad_conv_wgrad(h.ptr(xin), h.ptr(doutg), h.ptr(ws), h.ptr(dw32), h.ptr(db32), B, H, W, C, C, 9, st))
    wr = torch.zeros(C, C, 3, 3, requires_grad=True, dtype=torch.float64)
    br = torch.zeros(C, requires_grad=True, dtype=torch.float64)
    (F.conv2d(x.double(), wr, br, padding=1) * dout.double()).sum().backward()
    scale = wr.grad.abs().max().item()
    assert torch.allclose(dw.cpu().double(), wr.grad, atol=2e-4 * scale), (dw.cpu() - wr.grad).abs().max().item() / scale
    assert (dw - dw32).abs().max().item() <= 1e-5 * scale, (dw - dw32).abs().max().item() / scale
    assert torch.allclose(db.cpu().double(), br.grad, atol=2e-4 * br.grad.abs().max().item())
    e_h2 = float((dw.cpu().double() - wr.grad).pow(2).mean().sqrt())
    e_32 = float((dw32.cpu().double() - wr.grad).pow(2).mean().sqrt())
    if B * H * W >= 64:   # a real sum: the f32 accumulation dominates both kernels' errors
        assert e_h2 <= 1.5 * e_32 + 1e-12 * scale, (e_h2, e_32)
    else:                 # a handful of products: what is left is the operands' own 2^-22 (two planes of 11 bits; f32 itself: 2^-24)
        assert e_h2 <= 2.0 ** -22 * scale, (e_h2, e_32)
    dw2 = torch.zeros_like(dw)
    h.check(lib.lad_conv_wgrad_h2(h.ptr(xin), None, h.ptr(doutg), h.ptr(ws), h.ptr(dw2), None, B, H, W, C, st))
    assert torch.equal(dw2, dw)


def test_wgrad_h2_running_scales_follow_the_data():
    """A workgroup keeps one exponent per operand over its row range and lowers it when a tile no longer fits (re-basing the
    accumulators, re-staging the input window): magnitudes that GROW along the rows by 2^40 -- every few tiles another
    event -- and shrink again, all-zero regions first, and tiny gradients (1e-12) still give the float64 gradient."""
    h = _lib()
    lib = h.lib()
    st = h.stream_handle()
    C, B, H, W = 64, 24, 100, 44
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, H, W, generator=g)
    dout = torch.randn(B, C, H, W, generator=g) * 1e-12
    ramp = torch.linspace(-20, 20, H).view(1, 1, H, 1)          # 2^-20 ... 2^20 down every image
    x = x * torch.exp2(ramp)
    x[0:2] = 0.0                                                  # the first workgroups start on zeros
    dout = dout * torch.exp2(-ramp.flip(2) * 0.5)
    dout[5] *= 1e6
    xin, doutg = to_pnhwc(x), to_pnhwc(dout)
    ws = torch.zeros(int(lib.lad_conv_wgrad_workspace_floats(C, C, 9)), device="cuda")
    dw, db = torch.zeros(C, C, 3, 3, device="cuda"), torch.zeros(C, device="cuda")
    h.check(lib.lad_conv_wgrad_h2(h.ptr(xin), None, h.ptr(doutg), h.ptr(ws), h.ptr(dw), h.ptr(db), B, H, W, C, st))
    wr = torch.zeros(C, C, 3, 3, requires_grad=True, dtype=torch.float64)
    br = torch.zeros(C, requires_grad=True, dtype=torch.float64)
    (F.conv2d(x.double(), wr, br, padding=1) * dout.double()).sum().backward()
    scale = wr.grad.abs().max().item()
    assert bool(torch.isfinite(dw).all())
    assert float((dw.cpu().double() - wr.grad).abs().max()) <= 2e-6 * scale
    assert torch.allclose(db.cpu().double(), br.grad, atol=1e-5 * br.grad.abs().max().item())


@pytest.mark.parametrize("B,H,W", [(3, 13, 6), (29, 100, 44), (5, 7, 46), (1, 1, 1)])
def test_wgrad_h2_with_the_batchnorm_relu_applied_while_staging(B, H, W):
    """in_coef: bit-identical to lad_bn_act followed by the plain launch (lad_conv_wgrad_b3_bnrelu's contract)."""
    h = _lib()
    lib = h.lib()
    st = h.stream_handle()
    C = 64
    g = torch.Generator().manual_seed(B * 7 + H)
    rows, cnt = act_rows(B, H, W), B * H * W
    c1 = to_pnhwc(torch.randn(B, C, H, W, generator=g) * 2 + 1)
    dout = to_pnhwc(torch.randn(B, C, H, W, generator=g))
    gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.5 + 0.3).cuda()
    xn = from_pnhwc(c1, B, C, H, W).double()
    stat = torch.stack([xn.sum((0, 2, 3)), (xn ** 2).sum((0, 2, 3))]).float().reshape(-1).cuda()
    coef = torch.zeros(6 * C, device="cuda")
    h.check(lib.lad_bn_finalize(h.ptr(stat), 1, C, cnt, h.ptr(gam), h.ptr(bet), None, None, 0.1, h.ptr(coef), st))
    a1 = torch.zeros(rows * C, device="cuda")
    h.check(lib.lad_bn_act(h.ptr(c1), h.ptr(coef), None, None, h.ptr(a1), B, H, W, C, 1, st))
    ws = torch.zeros(int(lib.lad_conv_wgrad_workspace_floats(C, C, 9)), device="cuda")
    dw1, db1, dw2, db2 = (torch.zeros(n, device="cuda") for n in (C * C * 9, C, C * C * 9, C))
    h.check(lib.lad_conv_wgrad_h2(h.ptr(a1), None, h.ptr(dout), h.ptr(ws), h.ptr(dw1), h.ptr(db1), B, H, W, C, st))
    h.check(lib.lad_conv_wgrad_h2(h.ptr(c1), h.ptr(coef), h.ptr(dout), h.ptr(ws), h.ptr(dw2), h.ptr(db2), B, H, W, C, st))
    assert float(dw1.abs().max()) > 0 or cnt == 1
    assert torch.equal(dw1, dw2) and torch.equal(db1, db2)


@pytest.mark.parametrize("use_bits", [False, True])
@pytest.mark.parametrize("in_bn", [False, True])
@pytest.mark.parametrize("B,H,W", [(3, 13, 6), (29, 100, 44), (5, 7, 46), (1, 1, 1), (40, 50, 22), (2, 3, 45), (7, 2, 1), (1, 100, 44), (600, 4, 3)])
def test_wgrad_h2_with_the_batchnorm_backward_on_its_gradient_side(B, H, W, in_bn, use_bits):
    """lad_conv_wgrad_h2_bnbwd = lad_bn_bwd / lad_bn_bwd_bits (dx written) followed by lad_conv_wgrad_h2, bit for bit: the BatchNorm's
    input gradient dc it writes (zeros on border rows), the weight gradient and the bias gradient.  ReLU decisions recomputed
    from the BatchNorm's input or taken from sign bits; with and without the input-side BatchNorm + ReLU (in_coef)."""
    h = _lib()
    lib = h.lib()
    st = h.stream_handle()
    C = 64
    g = torch.Generator().manual_seed(B * 31 + H + 2 * int(in_bn) + int(use_bits))
    rows, cnt = act_rows(B, H, W), B * H * W

    def coef_of(t, gam, bet):
        xn = from_pnhwc(t, B, C, H, W).double()
        stat = torch.stack([xn.sum((0, 2, 3)), (xn ** 2).sum((0, 2, 3))]).float().reshape(-1).cuda()
        coef = torch.zeros(6 * C, device="cuda")
        h.check(lib.lad_bn_finalize(h.ptr(stat), 1, C, cnt, h.ptr(gam), h.ptr(bet), None, None, 0.1, h.ptr(coef), st))
        return coef

    xin = to_pnhwc(torch.randn(B, C, H, W, generator=g) * 1.5 + 0.5)           # the convolution's input (or its pre-BatchNorm form)
    in_coef = coef_of(xin, (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()) if in_bn else None
    cx = to_pnhwc(torch.randn(B, C, H, W, generator=g) * 2 + 1)                # the convolution's output = the BatchNorm's input
    dy = to_pnhwc(torch.randn(B, C, H, W, generator=g) * torch.exp(torch.randn(1, C, 1, 1, generator=g)))
    gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.4).cuda()
    coef = coef_of(cx, gam, bet)
    bits = None
    if use_bits:
        res = to_pnhwc(torch.randn(B, C, H, W, generator=g))
        y = torch.zeros(rows * C, device="cuda")
        bits = torch.zeros(rows, device="cuda", dtype=torch.int64)
        h.check(lib.lad_bn_act_bits(h.ptr(cx), h.ptr(coef), h.ptr(res), None, h.ptr(y), h.ptr(bits), B, H, W, C, st))
    bn_ws = torch.zeros(int(lib.lad_bn_bwd_workspace_floats(C)), device="cuda")
    ws = torch.zeros(int(lib.lad_conv_wgrad_workspace_floats(C, C, 9)), device="cuda")

    def bn_bwd(dx, bcoef, dg, db):
        if use_bits:
            h.check(lib.lad_bn_bwd_bits(h.ptr(dy), h.ptr(bits), h.ptr(cx), h.ptr(coef), h.ptr(gam), h.ptr(dx), h.ptr(dg), h.ptr(db),
                                        h.ptr(bn_ws), h.ptr(bcoef), None, 0, B, H, W, C, st), "lad_bn_bwd_bits")
        else:
            h.check(lib.lad_bn_bwd(h.ptr(dy), None, h.ptr(cx), h.ptr(coef), h.ptr(gam), None, None, None, h.ptr(dx), None, h.ptr(dg), h.ptr(db),
                                   None, None, h.ptr(bn_ws), h.ptr(bcoef), None, 0, B, H, W, C, 2, 0, st), "lad_bn_bwd")

    # the two launches
    dc1 = torch.full((rows * C,), 7.0, device="cuda")
    dc1[B * (H + 1) * (W + 1) * C:] = 0.0                        # (lad_bn_bwd leaves the tail rows alone: zero by the layout's invariant)
    bcoef1, dg1, db1 = torch.zeros(8 * C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    bn_bwd(dc1, bcoef1, dg1, db1)
    dw1, dbias1 = torch.zeros(C * C * 9, device="cuda"), torch.zeros(C, device="cuda")
    h.check(lib.lad_conv_wgrad_h2(h.ptr(xin), h.ptr(in_coef), h.ptr(dc1), h.ptr(ws), h.ptr(dw1), h.ptr(dbias1), B, H, W, C, st))
    # coefficients only, then the one launch
    dc2 = torch.full((rows * C,), -3.0, device="cuda")
    bcoef2, dg2, db2 = torch.zeros(8 * C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    bn_bwd(None, bcoef2, dg2, db2)
    assert torch.equal(bcoef1, bcoef2) and torch.equal(dg1, dg2) and torch.equal(db1, db2)
    dw2, dbias2 = torch.zeros(C * C * 9, device="cuda"), torch.zeros(C, device="cuda")
    h.check(lib.lad_conv_wgrad_h2_bnbwd(h.ptr(xin), h.ptr(in_coef), h.ptr(dy), h.ptr(cx), h.ptr(bits), h.ptr(coef), h.ptr(bcoef2), h.ptr(dc2),
                                        h.ptr(ws), h.ptr(dw2), h.ptr(dbias2), B, H, W, C, st), "lad_conv_wgrad_h2_bnbwd")
    body = B * (H + 1) * (W + 1) * C
    assert float(dc1[:body].abs().max()) > 0 or cnt == 1         # (one element: its BatchNorm gradient is exactly zero)
    assert torch.equal(dc1[:body], dc2[:body])
    assert float(dc2[body:].abs().max()) == 0.0                  # the tail rows of the layout are written as zeros
    assert borders_are_zero(dc2, B, C, H, W)
    assert torch.equal(dw1, dw2) and torch.equal(dbias1, dbias2)
    # argument checks
    assert lib.lad_conv_wgrad_h2_bnbwd(h.ptr(xin), None, h.ptr(dy), h.ptr(cx), None, h.ptr(coef), h.ptr(bcoef2), h.ptr(dy), h.ptr(ws), h.ptr(dw2),
                                       None, B, H, W, C, st) != 0
    assert lib.lad_conv_wgrad_h2_bnbwd(h.ptr(xin), None, h.ptr(dy), None, None, h.ptr(coef), h.ptr(bcoef2), h.ptr(dc2), h.ptr(ws), h.ptr(dw2),
                                       None, B, H, W, C, st) != 0
    assert lib.lad_conv_wgrad_h2_bnbwd(h.ptr(xin), None, h.ptr(dy), h.ptr(cx), None, h.ptr(coef), h.ptr(bcoef2), h.ptr(dc2), h.ptr(ws), h.ptr(dw2),
                                       None, B, H, W, 32, st) != 0


@pytest.mark.parametrize("B,seed", [(8, 41), (32, 42), (128, 43)])
def test_model_step_on_three_arithmetics(B, seed):
    """One train-mode forward + backward of the whole model on (a) the exact-f32 MFMA kernels, (b) three bf16 planes, (c) two f16
    planes, with the ReLU decisions of (a) imposed nowhere -- the three runs differ in rounding only, so apart from elements whose
    pre-activation sits within rounding of zero they agree: probabilities to 1e-5 (the oracle bar is 2e-5), every gradient tensor to a relative L2 of 2e-2 (the bar of the unmasked oracle comparisons)
    against (a), and (c) is as close to (a) as (b) is (within a factor of 3)."""
    from test_resnet_gpu import build_model, noise_grad, recipe
    xf = torch.from_numpy(recipe.make_features(seed, B)).cuda()
    tl = torch.from_numpy(recipe.make_labels(seed + 1, B)).cuda()
    res = {}
    for name, flags in (("f32", dict(bf16x3=False)), ("bf16x3", dict(bf16x3=True, f16x2=False)), ("f16x2", dict(bf16x3=True, f16x2=True))):
        m, _ = build_model(seed + 2)
        m.train()
        for k, v in flags.items():
            setattr(m.engine, k, v)
        p = m.engine.forward(xf, train=True, labels=tl).clone()
        m.engine.backward(None)
        res[name] = (p, {k: g.clone() for k, g in m.engine.grad_views().items()})
    p32, g32 = res["f32"]
    worst, pdiff = {}, {}
    for name in ("bf16x3", "f16x2"):
        p, g = res[name]
        pdiff[name] = float((p - p32).abs().max())
        assert pdiff[name] <= 1e-5, (name, pdiff[name])
        per = {}
        for k in g32:
            if noise_grad(k):
                continue
            l2 = float((g[k].double() - g32[k].double()).norm() / g32[k].double().norm())
            assert l2 <= 2e-2, (name, k, l2)   # (ReLU decisions at pre-activations within rounding of zero differ between runs)
            per[k] = l2
        worst[name] = per
    # "(c) as close to (a) as (b) is": on every tensor -- unless a ReLU decision fell differently in run (c) alone (a pre-activation within
    # rounding of zero; it moves every gradient BELOW it by 1e-4 .. 1e-2 and says nothing about the arithmetic: which run it happens in
    # changes with any rounding-level change upstream).  Then the comparison is made on the tensors no such decision touched in either run
    # (both below 1e-4), which must be a third of them at least (the layers above the decision).
    wf, wb = max(worst["f16x2"].values()), max(worst["bf16x3"].values())
    clean = [k for k in worst["f16x2"] if worst["f16x2"][k] < 1e-4 and worst["bf16x3"][k] < 1e-4]
    print(f"B={B}: max |p - p_f32| {pdiff}, worst gradient relative L2 vs f32 f16x2 {wf:.2e} bf16x3 {wb:.2e}, {len(clean)} of {len(worst['f16x2'])} tensors clean")
    if wf > 3.0 * wb + 1e-6:
        assert 3 * len(clean) >= len(worst["f16x2"]), (len(clean), wf, wb)
        cf, cb = max(worst["f16x2"][k] for k in clean), max(worst["bf16x3"][k] for k in clean)
        assert cf <= 3.0 * cb + 1e-6, (cf, cb)
    assert pdiff["f16x2"] <= 3.0 * pdiff["bf16x3"] + 1e-6, pdiff


@pytest.mark.parametrize("B,seed", [(8, 51), (64, 52)])
def test_model_gradient_with_the_batchnorm_backward_inside_the_weight_gradient(B, seed):
    """engine.fuse_bn_bwd_wgrad: the 64-channel BatchNorm backward's element-wise pass inside lad_conv_wgrad_h2_bnbwd (four launches per
    step) against the separate bn_bwd_apply launches -- same arithmetic, element for element: the whole flat gradient is bit-equal,
    and so are the intermediate gradients the debug capture sees."""
    from test_resnet_gpu import build_model, recipe
    xf = torch.from_numpy(recipe.make_features(seed, B)).cuda()
    tl = torch.from_numpy(recipe.make_labels(seed + 1, B)).cuda()
    res = {}
    for fuse in (False, True):
        m, _ = build_model(seed + 2)
        m.train()
        m.engine.fuse_bn_bwd_wgrad = fuse
        m.engine.debug_capture = {}
        p = m.engine.forward(xf, train=True, labels=tl).clone()
        m.engine.backward(None)
        res[fuse] = (p, m.engine.flat_grad().clone(), {k: {n: t for n, t in v.items() if t is not None} for k, v in m.engine.debug_capture.items()})
        m.engine.debug_capture = None
    assert torch.equal(res[False][0], res[True][0])
    assert float(res[True][1].abs().max()) > 0
    assert torch.equal(res[False][1], res[True][1])
    n_checked = 0
    for blk, d in res[False][2].items():
        for name, t in d.items():
            assert torch.equal(t, res[True][2][blk][name]), (blk, name)
            n_checked += 1
    assert n_checked >= 8


def test_one_table_packs_both_channel_counts_like_one_launch_each():
    """lad_conv_h2_pack_weights_multi with channels = 0 (every record names its own channel count: round 6, one pack launch per step)
    against one launch per channel count: identical images, both directions."""
    h = _lib()
    lib = h.lib()
    g = torch.Generator().manual_seed(77)
    ws = {C: (torch.randn(C, C, 3, 3, generator=g) * 0.2).cuda() for C in (64, 32)}
    recs, mixed, single = b"", {}, {}
    for C in (64, 32):
        for mode in (0, 1):
            single[(C, mode)] = pack_h2(h, lib, ws[C], mode, C)
            mixed[(C, mode)] = torch.zeros_like(single[(C, mode)])
            recs += struct.pack("<QQii", ws[C].data_ptr(), mixed[(C, mode)].data_ptr(), mode, C)
    table = torch.frombuffer(bytearray(recs), dtype=torch.uint8).cuda()
    h.check(lib.lad_conv_h2_pack_weights_multi(h.ptr(table), 4, 0, h.stream_handle()), "lad_conv_h2_pack_weights_multi (mixed)")
    torch.cuda.synchronize()
    for k in single:
        assert torch.equal(single[k], mixed[k]), k
