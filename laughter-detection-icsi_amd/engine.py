"""Launch sequencer for the ResNetBigger hot path on the MI355X (no autograd, no torch compute kernels).

`ResNetEngine` walks the layers of a `models.ResNetBigger` and issues the C-ABI calls of liblad_hip.so
(include/lad_hip.h) for
    forward            models.py:222-239 (ResNetBigger.forward), models.py:110-115 (ResidualBlock.forward)
    backward           what loss.backward() computes at train.py:289
    clip + Adam        train.py:291-295
on torch-owned device buffers.  PyTorch is used for memory, streams and (optionally) the dropout RNG only.

Memory plan.  Parameters live in ONE flat fp32 buffer in `model.parameters()` order
(each tensor 16-byte aligned), gradients in a second flat buffer of the same shape -> one all-reduce, one
norm, one Adam launch.  Activations are "PNHWC" with shared zero borders (csrc/lad_device.h): [batch][H+1][W+1][C] plus
a short tail; a plan for batch size B owns every activation / gradient buffer and is reused step after
step (no allocation inside the step).
"""
import ctypes
import math

import torch

import _hip

_VP = ctypes.c_void_p


# windows per launch group of predict_windows.  Half precision: the layers that still run per window (levels 3 and 4) are small
# launches, and 8192 windows amortise them better than 2048 did (2.03 -> 2.28 M windows/s on the 60 min channel); f32 keeps
# every window's level-1 activation (1.2 MB), so its groups stay at 2048
PREDICT_CHUNK = {"fp16": 8192, "fp32": 2048}
# fp16: windows whose level-1 / level-2 frame streams are computed in one go (engine.stream_super): five tensors of 5.8 KB per frame at
# level 1 -- 10 GB for a 60-minute channel's 360,000 windows, 30 GB at this cap
STREAM_SUPER_MAX = 1 << 20


class _LazyLevels(dict):
    """Rotating activation buffers of an eval plan, keyed by resolution level, created on first use."""

    def __init__(self, make):
        super().__init__()
        self._make = make

    def __missing__(self, key):
        value = self[key] = self._make(key)
        return value


def _align4(n):
    return (n + 3) & ~3


class _ConvSpec:
    """One convolution: geometry + where its weights live in the flat buffers."""

    def __init__(self, name, cin, cout, taps, stride, h_in, w_in, has_bias):
        self.name, self.cin, self.cout, self.taps, self.stride = name, cin, cout, taps, stride
        self.h_in, self.w_in = h_in, w_in
        self.h_out = (h_in + stride - 1) // stride
        self.w_out = (w_in + stride - 1) // stride
        self.has_bias = has_bias
        self.w = self.b = self.gw = self.gb = None  # views into the flat param / grad buffers
        self.wt_f = self.wt_d = None                # packed images (forward / data-gradient)


class _BnSpec:
    def __init__(self, name, c):
        self.name, self.c = name, c
        self.g = self.b = self.rm = self.rv = self.gg = self.gb = None
        self.coef = None  # float[6][C] scale, shift, mean, invstd, mean_lo, invstd_lo of the last forward


class _BlockSpec:
    def __init__(self, name):
        self.name = name
        self.conv1 = self.bn1 = self.conv2 = self.bn2 = self.sc_conv = self.sc_bn = None


class ResNetEngine:
    # flags that select kernels / fusions per layer: snapshotted by a train-mode forward, re-imposed during its backward
    KERNEL_OPTIONS = ("bf16x3", "bf16x3_32", "f16x2", "f16x2_32", "relu_bits", "virtual_a1", "fuse_bn_bwd", "fuse_bn_bwd_b3", "fuse_bn_bwd_wgrad", "fuse_s2_shortcut",
                      "fuse_s2_shortcut_wgrad", "s2_b3")

    def __init__(self, model):
        self.model = model
        self.device = None
        self._flat_p = self._flat_g = None
        self._plans = {}
        self._step_count = 0
        self._lib = None
        self._grad_dirty = False  # flat grad buffer holds a gradient that must be accumulated into
        self._train_forwards = 0
        self._fold_tag = None
        self._weights_version = 0
        # data-gradient epilogues can carry the first pass of the BatchNorm backward that follows (lad_conv_fwd_bnstat).
        # Measured at bs 512: -1.0 ms of reduce passes, +0.3 ms in the four fused conv launches, +0.3 ms in the finalize
        # kernels (18 k tile partials instead of 1 k): net -0.24 ms/step (0.9 %), while the dominant kernel's own launch
        # time grows 2.7 %.  Off by default for that reason; the path is covered by tests/test_resnet_gpu.py.
        self.fuse_bn_bwd = False
        self.overlap_wgrad = False  # weight gradients on a side stream (see _on_side); bench.py --overlap-wgrad
        # ... those of the 16- / 32-channel layers only: True, False, or "auto" = from 256 segments per step on (round 6, after the launch
        # merges: -0.65 % of the step at batch 512 in three A/B pairs on one box, 11.09 -> 11.01 ms; +3.5 % at batch 32, where every
        # kernel is launch-bound and the two event waits cost more than the overlap returns)
        self.overlap_wgrad_small = "auto"
        # the stem's batch statistics and its BatchNorm + weight-gradient backward from 54 moments of the input (one input channel): no
        # 64-channel statistics pass in forward, one pass over dy instead of two in backward (csrc/stem.hip; round 6)
        self.stem_onepass = True
        # The 64 -> 64 3x3 stride-1 convolutions (block1: 8 launches per step, forward + data gradient) run on the bf16 matrix
        # cores with three-way split operands (csrc/conv_b3.hip): fp32-equivalent results (2.9e-7 vs 4.4e-7 of the largest
        # output for the f32 MFMA, both against float64; tests/test_resnet_gpu.py) at 0.95 instead of 1.31 ms per launch.
        # False: every convolution on the exact-f32 MFMA.
        self.bf16x3 = True
        # ... and their data-gradient launches carry the first pass of the BatchNorm backward that consumes them (bn1 of the
        # block: mask recomputed from its input; bn2 of the block below: mask from its sign bits): three of the four
        # two-tensor reduce passes per step disappear for one tensor read in the epilogue.
        self.fuse_bn_bwd_b3 = True
        self.fuse_bn_bwd_wgrad = True   # the BatchNorm backward's element-wise pass inside the 64-channel weight-gradient launches
        # Round 4: the same layers on TWO f16 planes per operand instead of three bf16 planes (csrc/conv_h2.hip, wgrad_h2 in
        # csrc/wgrad_mfma.hip): three plane products per fp32-equivalent product instead of six, block floating point per staged
        # tile.  Error against float64 within 1.5x of the exact-f32 kernel's (tests/test_h2_gpu.py; the fp32 accumulation
        # dominates both), 0.78 -> 0.5 ms per convolution launch.  False: the bf16 x 3 kernels of rounds 2-3 (bf16x3 must be on).
        self.f16x2 = True
        self.f16x2_32 = True    # ... and block2's 32 -> 32 convolutions, forward and data gradient (their weight gradient stays bf16 x 3)
        self.bf16x3_32 = True   # block2's 32 -> 32 convolutions (forward, data gradient) on the same kernel: 0.116 -> 0.081 ms each
        # ... and the activation between the two convolutions of such a block stays virtual: BatchNorm + ReLU are applied
        # while conv2 and its weight gradient stage conv1's raw output (lad_conv_b3_fwd_f32_bnrelu, lad_conv_wgrad_b3_bnrelu).
        self.virtual_a1 = True
        # the 64 -> 32 stride-2 transition (forward + data gradient, with its shortcut) on the split-operand path: the space-to-depth
        # view of the input is formed while staging (csrc/conv_b3.hip, conv_s2b3 / dgrad_s2b3; round 3)
        self.s2_b3 = True
        # sliding-window inference, fp16: the stride-2 block behind level 1 reads the stream / strips directly (no assembled copy)
        self.stream_direct = True
        # ... and the SECOND resolution level is shared between the windows as well (two phase streams + strips)
        self.stream_level2 = True
        # ... and a 64-channel identity block on the boundary strips runs as ONE launch with the strip resident in LDS (round 5)
        self.strip_block_fused = True
        self.small_block_fused = True        # ... and the 16- / 32-channel identity blocks of small images likewise (several per workgroup)
        # fp16 sliding windows: the frame STREAMS of levels 1 and 2 are computed once for up to STREAM_SUPER_MAX windows, not once
        # per group of PREDICT_CHUNK windows (a group's stream launches are 1,500-tile launches: 11 % of its time); round 5
        self.stream_super = True
        # fp16 sliding windows: everything behind the shared level 2 (block3, block4, pooling, classifier) in one launch per group of
        # windows, a window resident in a CU's LDS throughout (lad_f16_tail_fwd; round 6) -- instead of nine launches of small kernels
        self.tail_fused = True
        # ... and block2.0's stride-2 entry on the level-2 strips reads its input from LDS (parity classes by LDS-DMA) instead of gathering it
        self.strip2_resident = True
        # ... and the strips have no stem launch: their inner rows ARE the stream's, the two edge rows are computed in the first block's launch
        self.strip_stem_shared = True
        self._tail_param_cache = {}
        self._sup_cache = {}
        self._sup_plans = {}                 # {"l1" / "l2": keys of the run-long eval plans, released with the run's buffer}
        self._probs_out = None               # (predict_windows: where the head of the current group of windows writes)
        # fp16 eval: a down-sampling block's 1x1 shortcut rides in its 3x3 convolution's launch (lad_f16_conv_s2_fwd*_sc; round 5)
        self.f16_s2_shortcut_fused = True
        self.fuse_s2_shortcut = True         # ... and its forward / data gradient inside conv1's launches (lad_conv_s2_*_fused)
        self.fuse_s2_shortcut_wgrad = True   # a stride-2 block's 1x1 shortcut weight gradient as a tenth tap of conv1's
        self.defer_wgrad_sums = True   # the 19 per-layer sums of weight-gradient slabs in one launch (csrc/slab_reduce.hip)
        self._defer_on = False
        self._cur_batch = 0
        self.relu_bits = True  # False: the residual ReLU mask is re-read from y and the shortcut gradient goes through HBM
        self._side = None
        self._side_readers = {}
        self._side_pending = False
        self.debug_capture = None  # tools/: dict that receives clones of the backward intermediates per block
        self.kernel_events = None  # bench.py: {kernel label: [(start_event, end_event), ...]} when profiling is on

    # ------------------------------------------------------------------------------------ flat storage
    def lib(self):
        if self._lib is None:
            self._lib = _hip.lib()
        return self._lib

    def _named_params(self):
        return list(self.model.named_parameters())

    def ensure_flat(self):
        """Move parameters into one flat buffer (and gradients into another) if they are not there already.
        Called at the top of every entry point: `model.to(device)` / `set_device` re-allocate parameter storage."""
        params = self._named_params()
        dev = params[0][1].device
        if dev.type != "cuda":
            raise _hip.LadHipError("ResNetBigger runs on the MI355X only: call model.set_device('cuda') first "
                                   "(the HIP path has no CPU fallback)")
        ok = self._flat_p is not None and self._flat_p.device == dev
        if ok:
            base = self._flat_p.data_ptr()
            for (name, p), off in zip(params, self._offsets):
                if p.data_ptr() != base + 4 * off or p.dtype != torch.float32:
                    ok = False
                    break
        if ok:
            return
        offs, total = [], 0
        for _, p in params:
            offs.append(total)
            total += _align4(p.numel())
        flat_p = torch.zeros(total, device=dev, dtype=torch.float32)
        flat_g = torch.zeros(total, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for (name, p), off in zip(params, offs):
                view = flat_p[off:off + p.numel()].view(p.shape)
                view.copy_(p.data.to(torch.float32))
                old_grad = p.grad
                p.data = view
                gview = flat_g[off:off + p.numel()].view(p.shape)
                if old_grad is not None:
                    gview.copy_(old_grad)
                    p.grad = gview
        self._flat_p, self._flat_g, self._offsets, self._n_flat = flat_p, flat_g, offs, total
        self._acc_g = None   # gradient-accumulation buffer of the fused loop (accumulated_grad)
        self._exp_avg = torch.zeros_like(flat_p)
        self._exp_avg_sq = torch.zeros_like(flat_p)
        self._norm_partials = torch.zeros(int(self.lib().lad_grad_sumsq_partials()), device=dev)
        self._norm_out = torch.zeros(1, device=dev)
        self._step_dev = torch.zeros(1, device=dev, dtype=torch.int64)
        self._rng_counter = torch.zeros(1, device=dev, dtype=torch.int64)   # dropout-mask draws so far (lad_head_fwd_train_rng)
        self._rng_seed = None
        self.device = dev
        self._plans = {}
        self._views = {name: (flat_p[off:off + p.numel()].view(p.shape), flat_g[off:off + p.numel()].view(p.shape))
                       for (name, p), off in zip(params, offs)}
        # the 22 `num_batches_tracked` counters become views of one int64 buffer: one increment launch per step
        bns = [m for m in self.model.modules() if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d))]
        self._nbt = torch.zeros(len(bns), device=dev, dtype=torch.int64)
        with torch.no_grad():
            for i, m in enumerate(bns):
                self._nbt[i] = m.num_batches_tracked.to(dev)
                m._buffers["num_batches_tracked"] = self._nbt[i]
        self._build_specs()
        self._weights_version += 1
        self._packed_version = {}
        self._pack_tables = {}
        self._param_list = [p for _, p in params]

    def bump_num_batches_tracked(self):
        self.ensure_flat()
        self._nbt.add_(1)

    def grad_views(self):
        return {k: v[1] for k, v in self._views.items()}

    def attach_grads(self):
        """Make every parameter's .grad the matching view of the flat gradient buffer."""
        for name, p in self._named_params():
            g = self._views[name][1]
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g

    # ------------------------------------------------------------------------------------ layer table
    def _build_specs(self):
        m = self.model
        bufs = dict(m.named_buffers())

        def conv(name, cin, cout, taps, stride, h, w, bias):
            s = _ConvSpec(name, cin, cout, taps, stride, h, w, bias)
            s.w, s.gw = self._views[name + ".weight"]
            if bias:
                s.b, s.gb = self._views[name + ".bias"]
            dev = self.device
            s.wt_f = torch.zeros(int(self.lib().lad_conv_packed_weight_floats(cout, cin, taps, 0)), device=dev)
            s.wt_d = torch.zeros(int(self.lib().lad_conv_packed_weight_floats(cout, cin, taps, 1)), device=dev)
            # b3: forward and data gradient on the split-operand kernel (64 or 32 channels); b3_full: 64 channels, where the
            # weight gradient, the sign bits and the virtual activation exist as well
            s.b3 = cin == cout and cin in (64, 32) and taps == 9 and stride == 1 and w <= 46
            s.b3_full = s.b3 and cin == 64
            s.b3_wgrad = s.b3 and (cin == 64 or w <= 30)   # the 32-channel weight-gradient window holds 64 rows + 2 (W + 2)
            if s.b3:  # split (bf16 x 3) weight images, forward and data gradient
                nb = int(self.lib().lad_conv_b3c_packed_weight_bytes(cin))
                s.wt3_f = torch.zeros(nb, device=dev, dtype=torch.uint8)
                s.wt3_d = torch.zeros(nb, device=dev, dtype=torch.uint8)
                nb2 = int(self.lib().lad_conv_h2_packed_weight_bytes(cin))   # ... and the f16 x 2 images (csrc/conv_h2.hip)
                s.wt2_f = torch.zeros(nb2, device=dev, dtype=torch.uint8)
                s.wt2_d = torch.zeros(nb2, device=dev, dtype=torch.uint8)
            # the 64 -> 32 stride-2 transition with its shortcut: one split image per direction (3x3 + 1x1 together)
            s.s2b3 = cin == 64 and cout == 32 and taps == 9 and stride == 2 and (w + 1) // 2 <= 45
            if s.s2b3:
                s.wt3_s2f = torch.zeros(int(self.lib().lad_conv_s2b3_packed_weight_bytes()), device=dev, dtype=torch.uint8)
                s.wt3_s2d = torch.zeros(int(self.lib().lad_conv_s2b3_dgrad_packed_weight_bytes()), device=dev, dtype=torch.uint8)
            return s

        def bn(name, c):
            s = _BnSpec(name, c)
            s.g, s.gg = self._views[name + ".weight"]
            s.b, s.gb = self._views[name + ".bias"]
            s.rm, s.rv = bufs[name + ".running_mean"], bufs[name + ".running_var"]
            if not (s.rm.is_cuda and s.rm.is_contiguous() and s.rm.dtype == torch.float32):
                raise _hip.LadHipError(f"{name}: running statistics must be contiguous float32 GPU tensors")
            return s

        self.stem_w, self.stem_gw = self._views["conv1.weight"]
        self.stem_cout = self.stem_w.shape[0]
        self.stem_bn = bn("bn1", self.stem_cout)
        self._block_defs = []
        cin = self.stem_cout
        for bi, cout in enumerate(m.filter_sizes, start=1):
            for j in range(2):
                stride = (1 if bi == 1 else 2) if j == 0 else 1
                self._block_defs.append((f"block{bi}.{j}", cin if j == 0 else cout, cout, stride))
            cin = cout
        self._conv_factory, self._bn_factory = conv, bn
        self._geom_specs = {}
        self.head_bn2 = bn("bn2", m.linear_layer_size)
        self.head_bn3 = bn("bn3", 32)
        names = ["bn2.weight", "bn2.bias", None, None, "linear1.weight", "linear1.bias", "bn3.weight", "bn3.bias",
                 None, None, "linear2.weight", "linear2.bias"]
        ptrs = []
        for i, n in enumerate(names):
            if n is not None:
                ptrs.append(self._views[n][0].data_ptr())
            else:
                t = {2: self.head_bn2.rm, 3: self.head_bn2.rv, 8: self.head_bn3.rm, 9: self.head_bn3.rv}[i]
                ptrs.append(t.data_ptr())
        self._head_params = (_VP * 12)(*ptrs)
        gnames = ["bn2.weight", "bn2.bias", "linear1.weight", "linear1.bias", "bn3.weight", "bn3.bias",
                  "linear2.weight", "linear2.bias"]
        self._head_grads = (_VP * 8)(*[self._views[n][1].data_ptr() for n in gnames])

    def _blocks_for(self, H, W, partial=False):
        """Layer specs for an (H, W) input (geometry-dependent: packed weights are shared, sizes are not).
        partial: only the leading full-resolution layers will run on this geometry (the stream / strip images of the
        sliding-window path): the pooling and classifier-size checks of a whole forward do not apply."""
        key = (H, W, partial)
        if key in self._geom_specs:
            return self._geom_specs[key]
        blocks = []
        h, w = H, W
        for name, cin, cout, stride in self._block_defs:
            b = _BlockSpec(name)
            b.conv1 = self._conv_factory(name + ".conv1", cin, cout, 9, stride, h, w, True)
            b.bn1 = self._bn_factory(name + ".bn1", cout)
            h2, w2 = b.conv1.h_out, b.conv1.w_out
            b.conv2 = self._conv_factory(name + ".conv2", cout, cout, 9, 1, h2, w2, True)
            b.bn2 = self._bn_factory(name + ".bn2", cout)
            if stride != 1 or cin != cout:
                b.sc_conv = self._conv_factory(name + ".shortcut.0", cin, cout, 1, stride, h, w, False)
                b.sc_bn = self._bn_factory(name + ".shortcut.1", cout)
            blocks.append(b)
            h, w = h2, w2
        if partial:
            self._geom_specs[key] = (blocks, h, w, 0)
            return self._geom_specs[key]
        if h < 4 or w < 4:
            raise ValueError(f"input ({H},{W}) is too small: AvgPool2d(4) sees a {h}x{w} map")
        feat = self._block_defs[-1][2] * (h // 4) * (w // 4)
        if feat != self.model.linear_layer_size:
            # same failure the reference hits (BatchNorm1d size check) for e.g. resnet_with_augmentation on (100,44)
            raise RuntimeError(f"running_mean should contain {feat} elements not {self.model.linear_layer_size}")
        self._geom_specs[key] = (blocks, h, w, feat)
        return self._geom_specs[key]

    # ------------------------------------------------------------------------------------ plans
    def _plan(self, B, H, W, train):
        key = (B, H, W, train)
        p = self._plans.get(key)
        if p is not None:
            return p
        dev = self.device
        blocks, h4, w4, feat = self._blocks_for(H, W)
        lib = self.lib()

        def act(h, w, c):
            return torch.zeros(int(lib.lad_act_rows(B, h, w)) * c, device=dev, dtype=torch.float32)

        p = {"blocks": blocks, "h4": h4, "w4": w4, "feat": feat}
        c0 = self.stem_cout
        p["stem_a"] = act(H, W, c0)
        p["stem_coef"] = torch.zeros(6 * c0, device=dev)
        if train:   # the stem's statistics and backward from moments of the input (stem_onepass)
            p["stem_mom"] = torch.zeros(int(lib.lad_stem_moments_doubles()), device=dev, dtype=torch.float64)
            p["stem_mom_ws"] = torch.zeros(int(lib.lad_stem_moments_workspace_doubles()), device=dev, dtype=torch.float64)
            p["stem_bwd_ws"] = torch.zeros(int(lib.lad_stem_bwd_onepass_workspace_floats()), device=dev)
        max_tiles = int(lib.lad_conv_num_tiles(B, H, W))
        for b in blocks:   # the stride-2 data gradient on the split-operand path writes its BatchNorm sums per parity class
            if getattr(b.conv1, "s2b3", False):
                max_tiles = max(max_tiles, int(lib.lad_conv_s2b3_dgrad_partials(B, b.conv1.h_in, b.conv1.w_in)))
        p["partials"] = torch.zeros(max_tiles * 2 * 64, device=dev)
        p["partials_sc"] = torch.zeros(max(int(lib.lad_conv_num_tiles(B, b.conv1.h_out, b.conv1.w_out)) * 2 * b.conv1.cout
                                           for b in blocks if b.sc_conv is not None) if any(b.sc_conv is not None for b in blocks) else 0,
                                       device=dev)
        acts = []
        for b in blocks:
            ho, wo, co = b.conv1.h_out, b.conv1.w_out, b.conv1.cout
            d = {"c1": act(ho, wo, co), "a1": act(ho, wo, co), "c2": act(ho, wo, co), "y": act(ho, wo, co),
                 "coef1": torch.zeros(6 * co, device=dev), "coef2": torch.zeros(6 * co, device=dev)}
            if b.sc_conv is not None:
                d["cs"] = act(ho, wo, co)
                d["coefs"] = torch.zeros(6 * co, device=dev)
            elif train and getattr(b.conv1, "b3_full", False):
                # sign bits of y, one uint64 per row: what the backward pass needs of the residual ReLU (lad_bn_act_bits)
                d["ybits"] = torch.zeros(int(lib.lad_act_rows(B, ho, wo)), device=dev, dtype=torch.int64)
            acts.append(d)
        p["acts"] = acts
        p["pooled"] = torch.zeros(B * feat, device=dev)
        p["probs"] = torch.zeros(B, device=dev)
        if train:
            p["h"] = torch.zeros(B * 32, device=dev)
            p["hstats"] = torch.zeros(2 * feat + 64, device=dev)
            p["metrics"] = torch.zeros(8, device=dev)
            p["head_ws"] = torch.zeros(int(lib.lad_head_workspace_floats(B, feat)), device=dev)
            p["dpooled"] = torch.zeros(B * feat, device=dev)
            # gradient scratch: per resolution level, 5 buffers sized for the widest tensor at that level
            levels = {}
            # buffers a (possibly side-stream) weight-gradient launch reads are never recycled inside one backward
            for b, d in zip(blocks, acts):
                ho, wo, co = b.conv1.h_out, b.conv1.w_out, b.conv1.cout
                d["dc1"], d["dc2"] = act(ho, wo, co), act(ho, wo, co)
                if b.conv1.stride != 1:
                    d["aux"] = act(ho, wo, co)  # gradient into the shortcut BatchNorm's input: read by its weight gradient
            sizes = {(H, W): c0}
            for b in blocks:
                sizes[(b.conv1.h_out, b.conv1.w_out)] = max(sizes.get((b.conv1.h_out, b.conv1.w_out), 0), b.conv1.cout)
                sizes[(b.conv1.h_in, b.conv1.w_in)] = max(sizes.get((b.conv1.h_in, b.conv1.w_in), 0), b.conv1.cin,
                                                            b.conv1.cout)
            for (h, w), c in sizes.items():
                levels[(h, w)] = [act(h, w, c) for _ in range(4)]
            p["g"] = levels
            ws = max(max(int(lib.lad_conv_wgrad_workspace_floats(cs.cin, cs.cout, cs.taps)),
                         int(lib.lad_conv_s2_wgrad_fused_workspace_floats(cs.cin, cs.cout)) if cs.stride != 1 else 0)
                     for b in blocks for cs in (b.conv1, b.conv2, b.sc_conv) if cs is not None)
            ws = max(ws, int(lib.lad_stem_wgrad_workspace_floats()), int(lib.lad_conv_wgrad_b3c_workspace_floats(32)))
            p["wgrad_ws"] = torch.zeros(ws, device=dev)
            # one workspace per convolution as well: with the slab sums deferred to one launch at the end of backward
            # (lad_wgrad_defer_*), every layer's partial slabs must survive until then (0.4 GB in all)
            p["wgrad_ws_of"] = {}
            for b in blocks:
                for cs in (b.conv1, b.conv2, b.sc_conv):
                    if cs is not None:
                        n = int(lib.lad_conv_wgrad_workspace_floats(cs.cin, cs.cout, cs.taps))
                        if getattr(cs, "b3_wgrad", False):
                            n = max(n, int(lib.lad_conv_wgrad_b3c_workspace_floats(cs.cin)))
                        if cs.stride != 1:   # (conv1 of a stride-2 block also holds the shortcut's slabs in the fused launch)
                            n = int(lib.lad_conv_s2_wgrad_fused_workspace_floats(cs.cin, cs.cout)) if cs.taps == 9 else \
                                int(lib.lad_conv_s2_wgrad_workspace_floats(cs.cin, cs.cout, cs.taps))
                        p["wgrad_ws_of"][cs.name] = torch.zeros(n, device=dev)
            p["bn_ws"] = torch.zeros(int(lib.lad_bn_bwd_workspace_floats(64)), device=dev)
            p["bcoef"] = torch.zeros(8 * 64, device=dev)
        self._plans[key] = p
        return p

    # ------------------------------------------------------------------------------------ helpers
    def _st(self):
        return _hip.stream_handle(self.device)

    def notify_weights_changed(self):
        """Parameters or running statistics were written behind the engine's back (a graph replay, a write through
        `param.data`, a broadcast): every cached derivative (packed MFMA images, BatchNorm folds, fp16 packs) is stale."""
        self._weights_version += 1

    def _pack_weights(self, blocks, need_dgrad):
        """Refresh the packed MFMA weight images if the parameters changed since the last pack."""
        # a Parameter's _version moves when torch writes it in place (optimizer.step, load_state_dict, init; after
        # `p.data = view` the Parameter keeps its OWN counter, the flat buffer's does not move);
        # _weights_version moves when our own Adam kernel writes the flat buffer
        ver = (self._weights_version, sum(p._version for p in self._param_list))
        arith = (self.bf16x3, self.f16x2, self.f16x2_32)
        tag = (ver, need_dgrad, arith)
        have = self._packed_version.get(id(blocks))
        if have == tag or have == (ver, True, arith):
            return
        lib, st = self.lib(), self._st()
        key = (id(blocks), need_dgrad)
        table = self._pack_tables.get(key)
        if table is None:  # device table of {w, wt, cout, cin, taps, mode} records: pointers never move
            import struct
            recs = b""
            n = 0
            for blk in blocks:
                for cs in (blk.conv1, blk.conv2, blk.sc_conv):
                    if cs is None:
                        continue
                    for mode in ((0, 1) if need_dgrad else (0,)):
                        wt = cs.wt_f if mode == 0 else cs.wt_d
                        recs += struct.pack("<QQiiii", cs.w.data_ptr(), wt.data_ptr(), cs.cout, cs.cin, cs.taps, mode)
                        n += 1
            dev_tab = torch.frombuffer(bytearray(recs), dtype=torch.uint8).to(self.device)
            table = self._pack_tables[key] = (dev_tab, n)
        _hip.check(lib.lad_conv_pack_weights_multi(_hip.ptr(table[0]), table[1], st), "lad_conv_pack_weights_multi")
        if self.bf16x3:
            for blk in blocks:
                if getattr(blk.conv1, "s2b3", False) and blk.sc_conv is not None:
                    c1 = blk.conv1
                    if need_dgrad:   # both images in one launch
                        _hip.check(lib.lad_conv_s2b3_pack_weights_pair(_hip.ptr(c1.w), _hip.ptr(blk.sc_conv.w), _hip.ptr(c1.wt3_s2f),
                                                                       _hip.ptr(c1.wt3_s2d), st), "lad_conv_s2b3_pack_weights_pair")
                    else:
                        _hip.check(lib.lad_conv_s2b3_pack_weights(_hip.ptr(c1.w), _hip.ptr(blk.sc_conv.w), _hip.ptr(c1.wt3_s2f), st),
                                   "lad_conv_s2b3_pack_weights")
                for cs in (blk.conv1, blk.conv2):
                    if cs.b3 and not self._h2(cs):
                        _hip.check(lib.lad_conv_b3c_pack_weights(_hip.ptr(cs.w), 0, _hip.ptr(cs.wt3_f), cs.cin, st), "lad_conv_b3c_pack_weights")
                        if need_dgrad:
                            _hip.check(lib.lad_conv_b3c_pack_weights(_hip.ptr(cs.w), 1, _hip.ptr(cs.wt3_d), cs.cin, st),
                                       "lad_conv_b3c_pack_weights")
            # the f16 x 2 images: ONE launch packs every layer and direction of both channel counts (a record names its own; 16 workgroups
            # per image)
            convs = [cs for blk in blocks for cs in (blk.conv1, blk.conv2) if cs.b3 and cs.cin in (64, 32) and self._h2(cs)]
            if convs:
                hkey = (id(blocks), need_dgrad, "h2")
                htab = self._pack_tables.get(hkey)
                if htab is None:
                    import struct
                    recs = b""
                    for cs in convs:
                        for mode in ((0, 1) if need_dgrad else (0,)):
                            recs += struct.pack("<QQii", cs.w.data_ptr(), (cs.wt2_f if mode == 0 else cs.wt2_d).data_ptr(), mode, cs.cin)
                    htab = self._pack_tables[hkey] = (torch.frombuffer(bytearray(recs), dtype=torch.uint8).to(self.device), len(recs) // 24)
                _hip.check(lib.lad_conv_h2_pack_weights_multi(_hip.ptr(htab[0]), htab[1], 0, st), "lad_conv_h2_pack_weights_multi")
        self._packed_version[id(blocks)] = tag

    def _mark(self, label):
        """HIP event on the launch stream (torch's current stream) when bench.py asked for per-kernel timing."""
        if self.kernel_events is None or label not in self.kernel_events:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(self.device))
        return ev

    def _mark_end(self, label, start):
        if start is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.device))
            self.kernel_events[label].append((start, ev))

    def _b3_fits(self, cs):
        # 64 channels (round 3): conv_b3x / wgrad_b3x address a tensor relative to the workgroup's own rows (64-bit bases),
        # only row NUMBERS are 32-bit.  32 channels: the weight gradient still runs on the round-2 kernel, whose byte offsets
        # are 32-bit (2 GiB per tensor = batch > 14,000 at 32 x 50 x 22); past that the layer runs on the exact-f32 kernels
        rows = self._cur_batch * (cs.h_in + 1) * (cs.w_in + 1) + cs.w_in + 2
        if cs.cin == 64:
            return rows < (1 << 31) - (1 << 20)
        return rows * cs.cin * 4 < (1 << 31) - (1 << 20)

    def _use_b3(self, cs):
        return self.bf16x3 and getattr(cs, "b3", False) and (cs.cin == 64 or self.bf16x3_32) and self._b3_fits(cs)

    def _h2(self, cs):
        """This split-operand layer runs on two f16 planes (csrc/conv_h2.hip) instead of three bf16 planes."""
        return self.f16x2 and getattr(cs, "b3", False) and (cs.cin == 64 or self.f16x2_32)

    def _conv_h2(self, cs, x, in_coef, wt, bias, addend, abits, out, partials, bn_x, bn_bits, bn_coef, B, h, w, what):
        _hip.check(self.lib().lad_conv_h2(_hip.ptr(x), _hip.ptr(in_coef), _hip.ptr(wt), _hip.ptr(bias), _hip.ptr(addend), _hip.ptr(abits),
                                          _hip.ptr(out), _hip.ptr(partials), _hip.ptr(bn_x), _hip.ptr(bn_bits), _hip.ptr(bn_coef),
                                          B, h, w, cs.cin, self._st()), f"lad_conv_h2({what}) {cs.name}")

    def _split_label(self, cs, transposed=False):
        """Kernel label of a split-operand launch for bench.py's per-kernel events (the data gradient swaps cin / cout)."""
        a, b = (cs.cout, cs.cin) if transposed else (cs.cin, cs.cout)
        return f"{'conv_h2' if self._h2(cs) else 'conv_b3'}<{a},{b},{cs.taps}>"

    def _use_b3_full(self, cs):
        return self.bf16x3 and getattr(cs, "b3_full", False) and self._b3_fits(cs)

    def _use_s2b3(self, b):
        c1 = b.conv1
        rows = self._cur_batch * (c1.h_in + 1) * (c1.w_in + 1) + c1.w_in + 2
        return (self.bf16x3 and self.s2_b3 and self.fuse_s2_shortcut and getattr(c1, "s2b3", False) and b.sc_conv is not None
                and rows < (1 << 31) - (1 << 20))

    def _use_bits(self, b, a):
        # identity-shortcut blocks on the split-operand kernels: the residual ReLU's decisions travel as sign bits
        # (8 bytes per row instead of re-reading y and writing / re-reading the masked gradient: csrc/bn.hip, conv_b3.hip)
        return self.relu_bits and "ybits" in a and b.sc_conv is None and self._use_b3_full(b.conv1) and not self.fuse_bn_bwd

    def _conv(self, cs, x, out, partials, B):
        lib, st = self.lib(), self._st()
        label = self._split_label(cs) if self._use_b3(cs) else f"conv_s{cs.stride}<{cs.cin},{cs.cout},{cs.taps}>"
        t0 = self._mark(label)
        self._conv_raw(cs, x, out, partials, B, lib, st)
        self._mark_end(label, t0)

    def _conv_raw(self, cs, x, out, partials, B, lib, st):
        if self._use_b3(cs) and self._h2(cs):
            self._conv_h2(cs, x, None, cs.wt2_f, cs.b, None, None, out, partials, None, None, None, B, cs.h_in, cs.w_in, "fwd")
        elif self._use_b3(cs):
            _hip.check(lib.lad_conv_b3c_fwd_f32(_hip.ptr(x), _hip.ptr(cs.wt3_f), _hip.ptr(cs.b), None, _hip.ptr(out),
                                                _hip.ptr(partials), B, cs.h_in, cs.w_in, cs.cin, st), "lad_conv_b3c_fwd_f32 " + cs.name)
        elif cs.stride == 1:
            _hip.check(lib.lad_conv_fwd(_hip.ptr(x), _hip.ptr(cs.wt_f), _hip.ptr(cs.b), None, _hip.ptr(out),
                                        _hip.ptr(partials), B, cs.h_in, cs.w_in, cs.cin, cs.cout, cs.taps, st),
                       "lad_conv_fwd " + cs.name)
        else:
            _hip.check(lib.lad_conv_s2_fwd(_hip.ptr(x), _hip.ptr(cs.wt_f), _hip.ptr(cs.b), _hip.ptr(out),
                                           _hip.ptr(partials), B, cs.h_in, cs.w_in, cs.cin, cs.cout, cs.taps, st),
                       "lad_conv_s2_fwd " + cs.name)

    def _bn_coef(self, bn, coef, partials, B, h, w, train):
        lib, st = self.lib(), self._st()
        if train:
            n_tiles = int(lib.lad_conv_num_tiles(B, h, w))
            _hip.check(lib.lad_bn_finalize(_hip.ptr(partials), n_tiles, bn.c, B * h * w, _hip.ptr(bn.g), _hip.ptr(bn.b),
                                           _hip.ptr(bn.rm), _hip.ptr(bn.rv), 0.1, _hip.ptr(coef), st),
                       "lad_bn_finalize " + bn.name)
        bn.coef = coef

    def _bn_coef_pair(self, bn_a, coef_a, part_a, bn_b, coef_b, part_b, B, h, w, train):
        """_bn_coef for a stride-2 block's bn1 and its shortcut BatchNorm (sums of one shape, left by one launch) in one launch."""
        if train:
            lib, st = self.lib(), self._st()
            n_tiles = int(lib.lad_conv_num_tiles(B, h, w))
            _hip.check(lib.lad_bn_finalize_pair(_hip.ptr(part_a), _hip.ptr(part_b), n_tiles, bn_a.c, B * h * w,
                                                _hip.ptr(bn_a.g), _hip.ptr(bn_a.b), _hip.ptr(bn_a.rm), _hip.ptr(bn_a.rv), _hip.ptr(coef_a),
                                                _hip.ptr(bn_b.g), _hip.ptr(bn_b.b), _hip.ptr(bn_b.rm), _hip.ptr(bn_b.rv), _hip.ptr(coef_b),
                                                0.1, st), "lad_bn_finalize_pair " + bn_a.name)
        bn_a.coef = coef_a
        bn_b.coef = coef_b

    def _bn_act(self, x, coef, res, rcoef, y, B, h, w, c, relu=1):
        _hip.check(self.lib().lad_bn_act(_hip.ptr(x), _hip.ptr(coef), _hip.ptr(res), _hip.ptr(rcoef), _hip.ptr(y), B, h, w, c,
                                         relu, self._st()), "lad_bn_act")

    # ------------------------------------------------------------------------------------ forward
    def forward(self, x, train, labels=None, drop_masks=None):
        """x: GPU float32 (B,1,H,W) or (B,H,W) -> probs (B,) (a plan-owned buffer, valid until the next call).

        train=True uses batch statistics, updates the running statistics and keeps what backward() needs.
        With labels (int32, (B,)) the head also produces the mean BCE loss and the metric counters.
        drop_masks: None (no dropout), a pair of mask tensors, or "rng": the head draws the masks of `model.dropout.p` itself (seeded
        by torch's CUDA generator seed) AND advances num_batches_tracked -- callers then skip bump_num_batches_tracked()."""
        self.ensure_flat()
        _hip.require_cuda(x, "x", torch.float32)
        if x.dim() == 4:
            if x.shape[1] != 1:
                raise ValueError("ResNetBigger expects a single input channel: (B,1,T,F)")
            B, _, H, W = x.shape
        elif x.dim() == 3:
            B, H, W = x.shape
        else:
            raise ValueError("x must be (B,1,T,F) or (B,T,F)")
        if B == 0:
            return torch.zeros(0, device=x.device)
        if not train:
            return self._forward_eval(x.view(-1), B, H, W, frame_stride=H, frames_avail=B * H)
        if train and B < 2:
            # torch: "Expected more than 1 value per channel when training" (BatchNorm1d on (1, F))
            raise ValueError("Expected more than 1 value per channel when training, got input size "
                             f"torch.Size([{B}, {self.model.linear_layer_size}])")
        lib, st = self.lib(), self._st()
        self._cur_batch = B
        p = self._plan(B, H, W, True)
        blocks = p["blocks"]
        self._pack_weights(blocks, need_dgrad=True)
        part = p["partials"]
        # stem (models.py:224)
        # The stem convolution (K = 9) is cheaper to recompute than to store: a statistics-only pass, then conv + BatchNorm +
        # ReLU in one kernel (the folded-BN stem kernel with the batch coefficients); the 596 MB convolution output is
        # never written, and backward() recomputes it the same way (lad_stem_bn_bwd_sums, lad_stem_wgrad_bn).
        if self.stem_onepass:
            # ... and its batch statistics need no convolution pass at all: one input channel, so sum x and sum x^2 are combinations of 54
            # moments of the nine taps (csrc/stem.hip, lad_stem_bn_stats: one pass over the 9 MB of features)
            bn = self.stem_bn
            _hip.check(lib.lad_stem_bn_stats(_hip.ptr(x), _hip.ptr(self.stem_w), _hip.ptr(bn.g), _hip.ptr(bn.b), _hip.ptr(bn.rm), _hip.ptr(bn.rv),
                                             0.1, _hip.ptr(p["stem_coef"]), _hip.ptr(p["stem_mom"]), _hip.ptr(p["stem_mom_ws"]), B, H, W,
                                             self.stem_cout, st), "lad_stem_bn_stats")
            bn.coef = p["stem_coef"]
            p["stem_mom_live"] = True
        else:
            _hip.check(lib.lad_stem_fwd(_hip.ptr(x), _hip.ptr(self.stem_w), None, _hip.ptr(part), B, H, W, self.stem_cout, st),
                       "lad_stem_fwd")
            self._bn_coef(self.stem_bn, p["stem_coef"], part, B, H, W, train)
            p["stem_mom_live"] = False
        c0 = self.stem_cout
        scale, shift = p["stem_coef"][:c0], p["stem_coef"][c0:2 * c0]
        _hip.check(lib.lad_stem_fwd_eval(_hip.ptr(x), _hip.ptr(self.stem_w), _hip.ptr(scale), _hip.ptr(shift), _hip.ptr(p["stem_a"]),
                                         B, H, W, c0, H, B * H, st), "lad_stem_fwd_eval (train)")
        cur = p["stem_a"]
        for b, a in zip(blocks, p["acts"]):
            ho, wo, co = b.conv1.h_out, b.conv1.w_out, b.conv1.cout
            fuse_sc_fwd = b.sc_conv is not None and b.conv1.stride != 1 and self.fuse_s2_shortcut
            if fuse_sc_fwd and train and self._use_s2b3(b):   # ... on the split-operand path (csrc/conv_b3.hip, conv_s2b3_kernel)
                label = f"conv_s2b3<{b.conv1.cin},{b.conv1.cout},9>"
                t0 = self._mark(label)
                _hip.check(lib.lad_conv_s2b3_fwd(_hip.ptr(cur), _hip.ptr(b.conv1.wt3_s2f), _hip.ptr(b.conv1.b), _hip.ptr(a["c1"]), _hip.ptr(part),
                                                 _hip.ptr(a["cs"]), _hip.ptr(p["partials_sc"]), B, b.conv1.h_in, b.conv1.w_in, st),
                           "lad_conv_s2b3_fwd " + b.conv1.name)
                self._mark_end(label, t0)
            elif fuse_sc_fwd:   # conv1 and the 1x1 shortcut convolution in one launch (csrc/conv_mfma.hip, conv_s2_kernel<SC>)
                label = f"conv_s2<{b.conv1.cin},{b.conv1.cout},9>"
                t0 = self._mark(label)
                _hip.check(lib.lad_conv_s2_fwd_fused(_hip.ptr(cur), _hip.ptr(b.conv1.wt_f), _hip.ptr(b.conv1.b), _hip.ptr(b.sc_conv.wt_f),
                                                     _hip.ptr(a["c1"]), _hip.ptr(part), _hip.ptr(a["cs"]), _hip.ptr(p["partials_sc"]),
                                                     B, b.conv1.h_in, b.conv1.w_in, b.conv1.cin, b.conv1.cout, st),
                           "lad_conv_s2_fwd_fused " + b.conv1.name)
                self._mark_end(label, t0)
            else:
                self._conv(b.conv1, cur, a["c1"], part, B)
            if fuse_sc_fwd:   # bn1 and the shortcut's BatchNorm: one launch
                self._bn_coef_pair(b.bn1, a["coef1"], part, b.sc_bn, a["coefs"], p["partials_sc"], B, ho, wo, train)
            else:
                self._bn_coef(b.bn1, a["coef1"], part, B, ho, wo, train)
            a["a1_virtual"] = self.virtual_a1 and self._use_b3(b.conv2) and getattr(b.conv2, "b3_wgrad", False)
            if a["a1_virtual"]:
                # relu(bn1(c1)) is formed while conv2 (and, in backward, its weight gradient) stage c1: never written
                label = self._split_label(b.conv2)
                t0 = self._mark(label)
                if self._h2(b.conv2):
                    self._conv_h2(b.conv2, a["c1"], a["coef1"], b.conv2.wt2_f, b.conv2.b, None, None, a["c2"], part, None, None, None,
                                  B, ho, wo, "fwd, bnrelu")
                else:
                    _hip.check(lib.lad_conv_b3c_fwd_f32_bnrelu(_hip.ptr(a["c1"]), _hip.ptr(a["coef1"]), _hip.ptr(b.conv2.wt3_f),
                                                               _hip.ptr(b.conv2.b), _hip.ptr(a["c2"]), _hip.ptr(part), B, ho, wo, b.conv2.cin, st),
                               "lad_conv_b3c_fwd_f32_bnrelu " + b.conv2.name)
                self._mark_end(label, t0)
            else:
                self._bn_act(a["c1"], a["coef1"], None, None, a["a1"], B, ho, wo, co)
                self._conv(b.conv2, a["a1"], a["c2"], part, B)
            self._bn_coef(b.bn2, a["coef2"], part, B, ho, wo, train)
            if b.sc_conv is not None:
                if not fuse_sc_fwd:
                    self._conv(b.sc_conv, cur, a["cs"], part, B)
                    self._bn_coef(b.sc_bn, a["coefs"], part, B, ho, wo, train)
                self._bn_act(a["c2"], a["coef2"], a["cs"], a["coefs"], a["y"], B, ho, wo, co)
            elif self._use_bits(b, a):
                _hip.check(lib.lad_bn_act_bits(_hip.ptr(a["c2"]), _hip.ptr(a["coef2"]), _hip.ptr(cur), None, _hip.ptr(a["y"]),
                                               _hip.ptr(a["ybits"]), B, ho, wo, co, st), "lad_bn_act_bits")
                a["bits_live"] = True
            else:
                self._bn_act(a["c2"], a["coef2"], cur, None, a["y"], B, ho, wo, co)
                a["bits_live"] = False
            a["x"] = cur
            cur = a["y"]
        last = blocks[-1].conv2
        _hip.check(lib.lad_pool_fwd(_hip.ptr(cur), _hip.ptr(p["pooled"]), B, p["h4"], p["w4"], last.cout, st), "lad_pool_fwd")
        if True:
            m1 = m2 = None
            rng = isinstance(drop_masks, str) and drop_masks == "rng"
            if drop_masks is not None and not rng:
                m1, m2 = drop_masks
                _hip.require_cuda(m1, "drop mask 1", torch.float32)
                _hip.require_cuda(m2, "drop mask 2", torch.float32)
                if tuple(m1.shape) != (B, p["feat"]) or tuple(m2.shape) != (B, 32):
                    raise ValueError("dropout masks must be (B,linear_layer_size) and (B,32)")
            if labels is not None:
                _hip.require_cuda(labels, "labels", torch.int32)
                if labels.numel() != B:
                    raise ValueError("labels must have one entry per sample")
            if rng:
                # the head's launch draws the two dropout masks itself and advances the BatchNorm layers' num_batches_tracked: none of
                # torch's four mask launches + one increment per step (csrc/head.hip, Philox4x32-10; round 6)
                keep = 1.0 - float(self.model.dropout.p)
                if keep < 1.0:
                    if "m1" not in p:
                        p["m1"] = torch.zeros((B, p["feat"]), device=self.device)
                        p["m2"] = torch.zeros((B, 32), device=self.device)
                    m1, m2 = p["m1"], p["m2"]
                seed = int(torch.cuda.default_generators[self.device.index or 0].initial_seed()) & ((1 << 64) - 1)
                if seed != self._rng_seed:       # (torch.manual_seed since the last draw: the sequence starts again)
                    self._rng_seed = seed
                    self._rng_counter.zero_()
                _hip.check(lib.lad_head_fwd_train_rng(self._head_params, _hip.ptr(p["pooled"]), B, p["feat"], _hip.ptr(m1), _hip.ptr(m2), keep,
                                                      seed, _hip.ptr(self._rng_counter), _hip.ptr(self._nbt), int(self._nbt.numel()),
                                                      _hip.ptr(labels), 0.1, _hip.ptr(p["h"]), _hip.ptr(p["hstats"]), _hip.ptr(p["probs"]),
                                                      _hip.ptr(p["metrics"]), st), "lad_head_fwd_train_rng")
            else:
                _hip.check(lib.lad_head_fwd_train(self._head_params, _hip.ptr(p["pooled"]), B, p["feat"], _hip.ptr(m1), _hip.ptr(m2),
                                                  _hip.ptr(labels), 0.1, _hip.ptr(p["h"]), _hip.ptr(p["hstats"]),
                                                  _hip.ptr(p["probs"]), _hip.ptr(p["metrics"]), st), "lad_head_fwd_train")
            p["saved"] = (x, labels, m1, m2, B, H, W)
            # the kernel choices of this forward pass: backward() must make the same ones (virtual activations that were
            # never written, sign bits that exist or not, packed weight images), whatever happens to the flags in between
            p["options"] = {k: getattr(self, k) for k in self.KERNEL_OPTIONS}
            self._last_train_plan = p
            self._train_forwards += 1  # running statistics moved: the eval-mode folds are stale
        return p["probs"]

    # ------------------------------------------------------------------------------------ eval (inference) path
    def _state_tag(self):
        bufs = sum(b._version for b in self.model.buffers())
        return (self._weights_version, sum(p._version for p in self._param_list), bufs, self._train_forwards)

    def _fold_eval(self, blocks):
        """Per-channel (scale, shift) of every BatchNorm that follows a convolution, from the running statistics
        (eval mode of models.py:110-115,224), refreshed only when parameters or statistics changed."""
        tags = self._fold_tag if isinstance(self._fold_tag, dict) else {}
        self._fold_tag = tags   # per layer table: the sliding-window path alternates between three geometries
        tag = self._state_tag()
        if tags.get(id(blocks)) == tag:
            return
        lib, st, dev = self.lib(), self._st(), self.device

        def fold(bn, conv_bias):
            if getattr(bn, "fold", None) is None:
                bn.fold = (torch.zeros(bn.c, device=dev), torch.zeros(bn.c, device=dev))
            _hip.check(lib.lad_bn_fold(_hip.ptr(bn.g), _hip.ptr(bn.b), _hip.ptr(bn.rm), _hip.ptr(bn.rv), _hip.ptr(conv_bias),
                                       bn.c, _hip.ptr(bn.fold[0]), _hip.ptr(bn.fold[1]), st), "lad_bn_fold " + bn.name)

        fold(self.stem_bn, None)
        for b in blocks:
            fold(b.bn1, b.conv1.b)
            fold(b.bn2, b.conv2.b)
            if b.sc_conv is not None:
                fold(b.sc_bn, None)
        tags[id(blocks)] = tag

    def _pack_f16(self, blocks):
        """Half-precision weight images for the fp16 inference kernels (refreshed with the folds)."""
        tags = getattr(self, "_f16_tags", None)
        if tags is None:
            tags = self._f16_tags = {}
        tag = self._state_tag()
        if tags.get(id(blocks)) == tag:
            return
        lib, st, dev = self.lib(), self._st(), self.device
        for b in blocks:
            for cs in (b.conv1, b.conv2, b.sc_conv):
                if cs is None:
                    continue
                if getattr(cs, "wt_h", None) is None:
                    cs.wt_h = torch.zeros(int(lib.lad_f16_packed_weight_halfs(cs.cout, cs.cin, cs.taps)), device=dev,
                                          dtype=torch.float16)
                _hip.check(lib.lad_f16_pack_weights(_hip.ptr(cs.w), cs.cout, cs.cin, cs.taps, _hip.ptr(cs.wt_h), st),
                           "lad_f16_pack_weights " + cs.name)
        tags[id(blocks)] = tag

    def _plan_eval(self, B, H, W, dtype=torch.float32, partial=False, owner=None):
        """owner: the run buffer ("l1" / "l2", _sup_buffer) this plan lives and dies with."""
        key = (B, H, W, "eval", dtype) + (("partial",) if partial else ())
        if owner is not None and key not in self._sup_plans.setdefault(owner, []):
            self._sup_plans[owner].append(key)
        p = self._plans.get(key)
        if p is not None:
            return p
        dev = self.device
        blocks, h4, w4, feat = self._blocks_for(H, W, partial)
        p = {"blocks": blocks, "h4": h4, "w4": w4, "feat": feat}
        levels = {(H, W): self.stem_cout}
        for b in blocks:
            k = (b.conv1.h_out, b.conv1.w_out)
            levels[k] = max(levels.get(k, 0), b.conv1.cout)
        # four rotating buffers per resolution level: block input, conv1 output, shortcut branch, block output
        rows_of = lambda k: int(self.lib().lad_act_rows(B, k[0], k[1]))  # noqa: E731
        # (allocated when a level is first used: the sliding-window path never touches the per-window buffers of the levels it
        # shares -- 19 GB for 8192 windows at level 1)
        p["lv"] = _LazyLevels(lambda k: [torch.zeros(rows_of(k) * levels[k], device=dev, dtype=dtype) for _ in range(4)])
        p["pooled"] = torch.zeros(B * feat, device=dev)
        p["probs"] = torch.zeros(B, device=dev)
        self._plans[key] = p
        return p

    def _conv_eval(self, cs, bn, x, addend, out, B, relu):
        lib, st = self.lib(), self._st()
        label = f"conv_s{cs.stride}<{cs.cin},{cs.cout},{cs.taps}>"
        t0 = self._mark(label)
        if cs.stride == 1:
            _hip.check(lib.lad_conv_fwd_eval(_hip.ptr(x), _hip.ptr(cs.wt_f), _hip.ptr(bn.fold[0]), _hip.ptr(bn.fold[1]),
                                             _hip.ptr(addend), _hip.ptr(out), B, cs.h_in, cs.w_in, cs.cin, cs.cout, cs.taps, relu, st),
                       "lad_conv_fwd_eval " + cs.name)
        else:
            if addend is not None:
                raise _hip.LadHipError("stride-2 eval convolution takes no residual")
            _hip.check(lib.lad_conv_s2_fwd_eval(_hip.ptr(x), _hip.ptr(cs.wt_f), _hip.ptr(bn.fold[0]), _hip.ptr(bn.fold[1]),
                                                _hip.ptr(out), B, cs.h_in, cs.w_in, cs.cin, cs.cout, cs.taps, relu, st),
                       "lad_conv_s2_fwd_eval " + cs.name)
        self._mark_end(label, t0)

    def _conv_eval_f16(self, cs, bn, x, addend, out, B, relu):
        lib, st = self.lib(), self._st()
        label = f"conv_f16_s{cs.stride}<{cs.cin},{cs.cout},{cs.taps}>"
        t0 = self._mark(label)
        if cs.stride == 1:
            _hip.check(lib.lad_f16_conv_fwd(_hip.ptr(x), _hip.ptr(cs.wt_h), _hip.ptr(bn.fold[0]), _hip.ptr(bn.fold[1]),
                                            _hip.ptr(addend), _hip.ptr(out), B, cs.h_in, cs.w_in, cs.cin, cs.cout, cs.taps, relu, st),
                       "lad_f16_conv_fwd " + cs.name)
        else:
            _hip.check(lib.lad_f16_conv_s2_fwd(_hip.ptr(x), _hip.ptr(cs.wt_h), _hip.ptr(bn.fold[0]), _hip.ptr(bn.fold[1]),
                                               _hip.ptr(out), B, cs.h_in, cs.w_in, cs.cin, cs.cout, cs.taps, relu, st),
                       "lad_f16_conv_s2_fwd " + cs.name)
        self._mark_end(label, t0)

    # ---- the eval forward in pieces (shared by the plain and the streaming sliding-window paths) ------------------------------
    def _eval_prepare(self, blocks, half):
        if half:
            self._fold_eval(blocks)
            self._pack_f16(blocks)
        else:
            self._pack_weights(blocks, need_dgrad=False)
            self._fold_eval(blocks)

    def _eval_stem(self, half, fptr, out, out_byte_offset, B, H, W, frame_stride, frames_avail):
        lib, st = self.lib(), self._st()
        optr = ctypes.c_void_p(out.data_ptr() + out_byte_offset)
        fn, name = (lib.lad_f16_stem_fwd, "lad_f16_stem_fwd") if half else (lib.lad_stem_fwd_eval, "lad_stem_fwd_eval")
        _hip.check(fn(fptr, _hip.ptr(self.stem_w), _hip.ptr(self.stem_bn.fold[0]), _hip.ptr(self.stem_bn.fold[1]), optr, B, H, W,
                      self.stem_cout, frame_stride, max(0, frames_avail), st), name)

    def _eval_blocks(self, half, p, blocks, cur, B, final_out=None):
        """Residual blocks `blocks` of plan p on the activation `cur` (one of the plan's rotating buffers of its level).
        final_out: where the LAST block's output goes instead of a rotating buffer (a slice of a caller's tensor)."""
        conv = self._conv_eval_f16 if half else self._conv_eval
        lv = p["lv"]
        for bi, b in enumerate(blocks):
            L = lv[(b.conv1.h_out, b.conv1.w_out)]
            free = [t for t in L if t is not cur]
            a1, y = free[0], free[1]
            if final_out is not None and bi == len(blocks) - 1:
                y = final_out
            if half and self._block_fits_lds(b, B):
                # both convolutions + the residual with the image(s) resident in LDS (csrc/conv_f16.hip: block_f16_strip_kernel at 64
                # channels, block_f16_small_kernel at 16 / 32)
                label = f"block_f16<{b.conv1.cin}>"
                t0 = self._mark(label)
                rc = self.lib().lad_f16_block_fwd(_hip.ptr(cur), _hip.ptr(b.conv1.wt_h), _hip.ptr(b.bn1.fold[0]), _hip.ptr(b.bn1.fold[1]),
                                                  _hip.ptr(b.conv2.wt_h), _hip.ptr(b.bn2.fold[0]), _hip.ptr(b.bn2.fold[1]), _hip.ptr(y),
                                                  B, b.conv1.h_in, b.conv1.w_in, b.conv1.cin, self._st())
                if rc != _hip.LAD_NOT_COVERED:      # (this geometry is not covered, nothing was launched -> the two convolutions)
                    _hip.check(rc, "lad_f16_block_fwd " + b.conv1.name)
                    self._mark_end(label, t0)
                    cur = y
                    continue
            if (half and b.sc_conv is not None and b.conv1.stride == 2 and b.sc_conv.stride == 2 and b.conv1.taps == 9
                    and b.sc_conv.taps == 1 and self._s2_shortcut_rides(b)):
                cs = free[2]
                label = f"conv_f16_s2sc<{b.conv1.cin},{b.conv1.cout}>"
                t0 = self._mark(label)
                _hip.check(self.lib().lad_f16_conv_s2_fwd_sc(
                    _hip.ptr(cur), _hip.ptr(b.conv1.wt_h), _hip.ptr(b.bn1.fold[0]), _hip.ptr(b.bn1.fold[1]), _hip.ptr(a1),
                    _hip.ptr(b.sc_conv.wt_h), _hip.ptr(b.sc_bn.fold[0]), _hip.ptr(b.sc_bn.fold[1]), _hip.ptr(cs), B, b.conv1.h_in,
                    b.conv1.w_in, b.conv1.cin, b.conv1.cout, 1, self._st()), "lad_f16_conv_s2_fwd_sc " + b.conv1.name)
                self._mark_end(label, t0)
                conv(b.conv2, b.bn2, a1, cs, y, B, 1)
                cur = y
                continue
            conv(b.conv1, b.bn1, cur, None, a1, B, 1)
            if b.sc_conv is not None:
                cs = free[2]
                conv(b.sc_conv, b.sc_bn, cur, None, cs, B, 0)
                conv(b.conv2, b.bn2, a1, cs, y, B, 1)
            else:
                conv(b.conv2, b.bn2, a1, cur, y, B, 1)
            cur = y
        return cur

    def _s2_shortcut_rides(self, b):
        """fp16 eval: the 1x1 shortcut of down-sampling block b inside conv1's launch?  Measured per group of 8,192 windows
        (profiles/r05_infer_s2_shortcut.log): 64 -> 32: 344 us against 314 + 72; 16 -> 16: 41 against 30 + 14; 32 -> 16: 219 against
        157 + 44 -- there the second accumulator costs the launch more than the second gather saves."""
        return self.f16_s2_shortcut_fused and b.conv1.cin != 32

    def _block_fits_lds(self, b, B):
        """Worth trying lad_f16_block_fwd (include/lad_hip.h)?  Identity block; 64 channels on >= 256 images of at most 512 positions
        (the boundary strips of level 1), or 16 / 32 channels on >= 512 small images (the strips of level 2, the windows at levels
        3 and 4).  The entry point itself answers LAD_NOT_COVERED for what does not fit a CU's LDS."""
        c = b.conv1
        if b.sc_conv is not None or c.stride != 1 or c.taps != 9 or c.cin != c.cout:
            return False
        img = (c.h_in + 1) * (c.w_in + 1)
        if c.cin == 64:
            return self.strip_block_fused and B >= 256 and img <= 512 and img + c.w_in <= 562
        return self.small_block_fused and c.cin in (16, 32) and B >= 512 and img <= 2048

    def _eval_tail(self, half, p, cur, B):
        lib, st = self.lib(), self._st()
        last = p["blocks"][-1].conv2
        p["block_out"] = cur
        pool, name = (lib.lad_f16_pool_fwd, "lad_f16_pool_fwd") if half else (lib.lad_pool_fwd, "lad_pool_fwd")
        _hip.check(pool(_hip.ptr(cur), _hip.ptr(p["pooled"]), B, p["h4"], p["w4"], last.cout, st), name)
        # predict_windows hands in the slice of ITS output these windows belong to: the head writes there (no copy per chunk)
        probs = self._probs_out if self._probs_out is not None else p["probs"]
        _hip.check(lib.lad_head_fwd_eval(self._head_params, _hip.ptr(p["pooled"]), B, p["feat"], _hip.ptr(probs), st),
                   "lad_head_fwd_eval")
        return probs

    def _forward_eval_any(self, half, feat_flat, B, H, W, frame_stride, frames_avail, feat_offset_floats=0):
        """Eval-mode forward of B images taken from a (frames, W) feature matrix (see lad_stem_fwd_eval): every BatchNorm is
        folded into the epilogue of the convolution in front of it, so the whole model is stem + 19 convolution launches +
        pool + head.  half: activations / weights in fp16 on the 16-bit matrix cores (csrc/conv_f16.hip), f32 in and out."""
        p = self._plan_eval(B, H, W, torch.float16 if half else torch.float32)
        blocks = p["blocks"]
        self._eval_prepare(blocks, half)
        cur = p["lv"][(H, W)][0]
        fptr = ctypes.c_void_p(feat_flat.data_ptr() + 4 * feat_offset_floats)
        self._eval_stem(half, fptr, cur, 0, B, H, W, frame_stride, frames_avail)
        cur = self._eval_blocks(half, p, blocks, cur, B)
        return self._eval_tail(half, p, cur, B)

    def _forward_eval_f16(self, feat_flat, B, H, W, frame_stride, frames_avail, feat_offset_floats=0):
        return self._forward_eval_any(True, feat_flat, B, H, W, frame_stride, frames_avail, feat_offset_floats)

    def _forward_eval(self, feat_flat, B, H, W, frame_stride, frames_avail, feat_offset_floats=0):
        return self._forward_eval_any(False, feat_flat, B, H, W, frame_stride, frames_avail, feat_offset_floats)

    def _forward_eval_stream(self, half, feat_flat, B, H, W, frames_avail, feat_offset_floats=0, sup=None):
        """The same probabilities for B windows AT A STRIDE OF ONE FRAME, with the full-resolution layers (stem + the stride-1
        blocks of level 1: 75 % of the model's arithmetic) run once over the shared stream and on one boundary strip per
        frame offset instead of on every window (csrc/gather.hip, lad_assemble_windows, for the argument): a ninth of that work.
        In half precision the second level is shared the same way (_eval_level2_shared).
        sup (predict_windows, fp16): the streams are computed ONCE for a run of groups -- {"d": this group's first window within the
        run, "S": windows of the run, "B_max": its largest group, "base": address of the run's first frame, "frames_avail": frames
        from there on} + what this function keeps in it.  A row of a stream depends on the frames within `band` rows of it only, so
        every row a window uses is the row the group's own stream image would hold, bit for bit."""
        dtype = torch.float16 if half else torch.float32
        pw = self._plan_eval(B, H, W, dtype)
        blocks = pw["blocks"]
        n1 = 0
        while n1 < len(blocks) and blocks[n1].conv1.stride == 1 and blocks[n1].sc_conv is None and blocks[n1].conv1.h_out == H:
            n1 += 1
        band = 1 + 2 * n1                       # 3x3 convolutions at full resolution: the stem + two per block
        if n1 == 0 or H < 4 * band or B < 2:    # nothing to share
            return self._forward_eval_any(half, feat_flat, B, H, W, 1, frames_avail, feat_offset_floats)
        lib, st = self.lib(), self._st()
        Hs, Ht = B + H - 1, 2 * band
        ps = self._plan_eval(1, Hs, W, dtype, partial=True)
        n_strip = B + H - Ht                    # one strip per frame offset: the top rows of one window, the bottom rows of another
        pt = self._plan_eval(n_strip, Ht, W, dtype, partial=True)
        for p in (pw, ps, pt):
            self._eval_prepare(p["blocks"], half)
        esize = 2 if half else 4
        C = self.stem_cout
        base = feat_flat.data_ptr() + 4 * feat_offset_floats
        nb = blocks[n1] if n1 < len(blocks) else None
        direct = (half and nb is not None and nb.sc_conv is not None and nb.conv1.stride == 2
                  and (nb.conv1.cin, nb.conv1.cout) == (64, 32) and self.stream_direct)
        # `direct`: the strips and the stream go into ONE buffer and the stride-2 block that follows reads every window's rows
        # from where they lie (lad_f16_conv_s2_fwd_windows) -- no assembled copy (1.2 GB written and read per 2048 windows)
        img_t_rows = (Ht + 1) * (W + 1)
        cat, out_t, out_s = None, None, None
        # level 2 shared as well: the stride-2 block and the stride-1 blocks behind it, up to the next stride-2 block
        k3 = n1 + 1
        while direct and k3 < len(blocks) and blocks[k3].conv1.stride == 1 and blocks[k3].sc_conv is None:
            k3 += 1
        n2 = k3 - n1 - 1
        margin2 = 1 + 2 * n2                    # stride-1 3x3 convolutions at level 2
        band2 = band // 2 + 1 + margin2         # rows of a window that differ from the stream at the end of level 2 (either end)
        Ht2 = 2 * band2                         # level-2 strips, paired like level 1's: top of window s over bottom of window s - shift2
        shift2 = 2 * (H // 2 - Ht2)
        share2 = (direct and self.stream_level2 and H % 2 == 0 and k3 < len(blocks) and blocks[k3].sc_conv is not None
                  and blocks[k3].conv1.stride == 2 and (blocks[k3].conv1.cin, blocks[k3].conv1.cout) == (32, 16)
                  and H // 2 >= 2 * Ht2)
        if sup is not None and not share2:
            sup = None                          # (the run's streams are a feature of the fully shared path)
        if sup is not None:
            # ONE buffer for the run: [strips of the current group (room for the largest)][the run's stream]; a group's windows find
            # their stream rows sup["d"] rows further down
            n_strip_max, Hs_S = sup["B_max"] + H - Ht, sup["S"] + H - 1
            sup["stream_row0"] = n_strip_max * img_t_rows
            n_rows = n_strip_max * img_t_rows + (Hs_S + 1) * (W + 1) + W + 2 + 2 * (W + 1)
            cat = sup.get("cat")
            if cat is None:
                cat = sup["cat"] = self._sup_buffer("l1", n_rows * C, dtype, (sup["S"], sup["B_max"], H, W))
                psS = self._plan_eval(1, Hs_S, W, dtype, partial=True, owner="l1")
                self._eval_prepare(psS["blocks"], half)
                if half and self.strip_stem_shared:
                    # the run's stem output is KEPT (2 GB for a 60-minute channel): the strips' first block reads its inner rows from it
                    cS = sup["stem"] = self._sup_buffer("l0", int(lib.lad_act_rows(1, Hs_S, W)) * C, dtype, (sup["S"], sup["B_max"], H, W))
                else:
                    cS = psS["lv"][(Hs_S, W)][0]
                self._eval_stem(half, ctypes.c_void_p(sup["base"]), cS, 0, 1, Hs_S, W, 1, sup["frames_avail"])
                self._eval_blocks(half, psS, psS["blocks"][:n1], cS, 1, final_out=cat[sup["stream_row0"] * C:])
            out_t = cat[:(n_strip * img_t_rows + W + 2) * C]
            if n_strip < n_strip_max:           # (behind a shorter last group's strips lie an earlier group's: the W + 2 rows its last
                cat[n_strip * img_t_rows * C:(n_strip * img_t_rows + W + 2) * C].zero_()   # strip reads below itself must be zero)
        elif direct:
            # (+ two zero rows: the odd-phase level-2 stream reads the level-1 stream from its second row on)
            n_rows = n_strip * img_t_rows + (Hs + 1) * (W + 1) + W + 2 + (2 * (W + 1) if share2 else 0)
            cat = pw.get("l1cat")
            if cat is None or cat.numel() != n_rows * C:
                cat = pw["l1cat"] = torch.zeros(n_rows * C, device=self.device, dtype=dtype)
            out_t = cat[:(n_strip * img_t_rows + W + 2) * C]    # (the strips' tail rows are the stream's border row: zeros either way)
            out_s = cat[n_strip * img_t_rows * C:]
        stem_keep, stem_rows, stem_row0 = (sup.get("stem"), sup["S"] + H - 1, sup["d"]) if sup is not None else (None, Hs, 0)
        if sup is None:
            # the stream: frames [0, B + H - 1) of the chunk as one tall image
            if half and self.strip_stem_shared:
                if ps.get("stem_keep") is None:
                    ps["stem_keep"] = torch.zeros(int(lib.lad_act_rows(1, Hs, W)) * C, device=self.device, dtype=dtype)
                cs_ = stem_keep = ps["stem_keep"]
            else:
                cs_ = ps["lv"][(Hs, W)][0]
            self._eval_stem(half, ctypes.c_void_p(base), cs_, 0, 1, Hs, W, 1, frames_avail)
            cs_ = self._eval_blocks(half, ps, ps["blocks"][:n1], cs_, 1, final_out=out_s)
        # the strips: frames [s, s + 2 band) for every offset s (gather.hip: upper half = top of window s, lower half = bottom of
        # window s - (H - 2 band))
        ct = None
        b0 = pt["blocks"][0]
        if half and self.strip_stem_shared and stem_keep is not None and Ht >= 3 and self._block_fits_lds(b0, n_strip) and b0.conv1.cin == 64:
            # Rows 1 .. Ht - 2 of strip s ARE rows s + 1 .. of the stream's stem output (their input frames lie inside the strip either way):
            # the first block's launch takes them from there and computes the strip's first and last row itself
            # (lad_f16_block_fwd_stem_rows; round 6) -- no stem launch for the strips, no strip-sized stem tensor written or read
            y0 = out_t if n1 == 1 else pt["lv"][(Ht, W)][1]
            label = f"block_f16<{b0.conv1.cin}>"
            t0 = self._mark(label)
            rc = lib.lad_f16_block_fwd_stem_rows(_hip.ptr(stem_keep), stem_rows, stem_row0, ctypes.c_void_p(base), max(0, frames_avail),
                                                 _hip.ptr(self.stem_w), _hip.ptr(self.stem_bn.fold[0]), _hip.ptr(self.stem_bn.fold[1]),
                                                 _hip.ptr(b0.conv1.wt_h), _hip.ptr(b0.bn1.fold[0]), _hip.ptr(b0.bn1.fold[1]),
                                                 _hip.ptr(b0.conv2.wt_h), _hip.ptr(b0.bn2.fold[0]), _hip.ptr(b0.bn2.fold[1]), _hip.ptr(y0),
                                                 n_strip, Ht, W, st)
            if rc != _hip.LAD_NOT_COVERED:
                _hip.check(rc, "lad_f16_block_fwd_stem_rows " + b0.conv1.name)
                self._mark_end(label, t0)
                ct = y0 if n1 == 1 else self._eval_blocks(half, pt, pt["blocks"][1:n1], y0, n_strip, final_out=out_t)
        if ct is None:
            ct = pt["lv"][(Ht, W)][0]
            self._eval_stem(half, ctypes.c_void_p(base), ct, 0, n_strip, Ht, W, 1, frames_avail)
            ct = self._eval_blocks(half, pt, pt["blocks"][:n1], ct, n_strip, final_out=out_t)
        if share2:
            return self._eval_level2_shared(pw, cat, n_rows, n_strip, B, H, W, band, Ht, Hs, n1, k3, band2, Ht2, shift2, sup)
        if direct:
            L = pw["lv"][(nb.conv1.h_out, nb.conv1.w_out)]
            a1, cs2, y = L[0], L[1], L[2]
            if self._s2_shortcut_rides(nb):
                label = f"conv_f16_s2sc<{nb.conv1.cin},{nb.conv1.cout}>"
                t0 = self._mark(label)
                _hip.check(lib.lad_f16_conv_s2_fwd_mapped_sc(
                    _hip.ptr(cat), _hip.ptr(nb.conv1.wt_h), _hip.ptr(nb.bn1.fold[0]), _hip.ptr(nb.bn1.fold[1]), _hip.ptr(a1),
                    _hip.ptr(nb.sc_conv.wt_h), _hip.ptr(nb.sc_bn.fold[0]), _hip.ptr(nb.sc_bn.fold[1]), _hip.ptr(cs2), B, H, W, band, Ht,
                    H - Ht, n_strip * img_t_rows, 1, 0, n_rows, 0, nb.conv1.cin, nb.conv1.cout, 1, st),
                    "lad_f16_conv_s2_fwd_mapped_sc " + nb.conv1.name)
                self._mark_end(label, t0)
            else:
                for cs_spec, bn, dst, relu in ((nb.conv1, nb.bn1, a1, 1), (nb.sc_conv, nb.sc_bn, cs2, 0)):
                    label = f"conv_f16_s2<{cs_spec.cin},{cs_spec.cout},{cs_spec.taps}>"
                    t0 = self._mark(label)
                    _hip.check(lib.lad_f16_conv_s2_fwd_windows(_hip.ptr(cat), _hip.ptr(cs_spec.wt_h), _hip.ptr(bn.fold[0]), _hip.ptr(bn.fold[1]),
                                                               _hip.ptr(dst), B, H, W, band, cs_spec.cin, cs_spec.cout, cs_spec.taps, relu, st),
                               "lad_f16_conv_s2_fwd_windows " + cs_spec.name)
                    self._mark_end(label, t0)
            self._conv_eval_f16(nb.conv2, nb.bn2, a1, cs2, y, B, 1)
            cur = self._eval_blocks(half, pw, blocks[n1 + 1:], y, B)
            return self._eval_tail(half, pw, cur, B)
        # every window's level-1 output, then the rest of the model per window
        cur = pw["lv"][(H, W)][0]
        _hip.check(lib.lad_assemble_windows(_hip.ptr(cs_), _hip.ptr(ct), _hip.ptr(cur), B, H, W, band, C * esize, st), "lad_assemble_windows")
        cur = self._eval_blocks(half, pw, blocks[n1:], cur, B)
        return self._eval_tail(half, pw, cur, B)

    def _sup_buffer(self, name, numel, dtype, layout):
        """Zero-initialised buffer of a run's streams (+ the strips of its current group), kept between runs of the same LAYOUT
        (run length, largest group, window geometry: [strips][stream][spare zero rows] -- two runs of equal size but another split
        would find old stream data where zero rows are expected): every region of it is rewritten or explicitly zeroed where a reader
        expects zeros (the W + 2 rows behind a shorter last group's strips); the border rows, tails and spare rows that nobody writes
        stay as allocated.  A run of another layout REPLACES the buffer and the run-long eval plans that fed it (four activation
        buffers per level: 8 + 2 GB for a 60-minute channel), so a long-lived process holds one run's memory whatever it predicts."""
        key = ("sup", name, numel, dtype, layout)
        buf = self._sup_cache.get(key)
        if buf is None:
            for k in [k for k in self._sup_cache if k[1] == name]:
                del self._sup_cache[k]
            for pk in self._sup_plans.pop(name, []):
                self._release_plan(pk)
            buf = self._sup_cache[key] = torch.zeros(numel, device=self.device, dtype=dtype)
        return buf

    def _release_plan(self, key):
        """Forget an eval plan and, when no other plan shares its layer table, everything cached per table (geometry specs with their
        packed images, fold / pack tags -- keyed by id(), which a later table may reuse)."""
        p = self._plans.pop(key, None)
        if p is None:
            return
        blocks = p["blocks"]
        if any(q.get("blocks") is blocks for q in self._plans.values()):
            return
        for gk in [gk for gk, gv in self._geom_specs.items() if gv[0] is blocks]:
            del self._geom_specs[gk]
        for tags in (self._fold_tag if isinstance(self._fold_tag, dict) else {}, getattr(self, "_f16_tags", None) or {}, self._packed_version):
            tags.pop(id(blocks), None)
        for k in [k for k in self._pack_tables if k[0] == id(blocks)]:
            del self._pack_tables[k]

    def _eval_level2_shared(self, pw, cat, cat_rows, n_strip, B, H, W, band, Ht, Hs, n1, k3, band2, Ht2, shift2, sup=None):
        """fp16 sliding windows, second resolution level.  Row r of window i at level 2 looks at level-1 rows 2r - 1 .. 2r + 1 of the
        window = stream rows i + 2r - 1 ..: windows i = 2j + phase share ONE level-2 stream per phase (row j + r of it), which
        is the stride-2 block run on the level-1 stream from row `phase` on; the rows that see a window's own top / bottom
        (band2 of them, either end) come from strips of Ht2 = 2 band2 rows, paired like level 1's (strip s = the first band2
        rows of window s over the last band2 of window s - shift2: the same 12 positions of the same phase stream, padded
        above for the one and below for the other), whose stride-2 layer reads the level-1 strips and stream through the
        window map.  The stride-2 block of level 3 then reads every window from the two streams and the strips
        (lad_f16_conv_s2_fwd_mapped, phases = 2), and the rest of the model runs per window as before."""
        lib, st = self.lib(), self._st()
        dtype, esize = torch.float16, 2
        blocks = pw["blocks"]
        nb = blocks[n1]
        C1, C2 = nb.conv1.cin, nb.conv1.cout
        H2, W2 = nb.conv1.h_out, nb.conv1.w_out
        Wp, Wp2 = W + 1, W2 + 1
        Hs_S = Hs if sup is None else sup["S"] + H - 1             # rows of the level-1 stream image (the group's / the run's)
        h2s = (Hs_S + 1) // 2                                      # rows of a level-2 stream image
        ps2 = self._plan_eval(2, 2 * h2s, W, dtype, partial=True)
        n_strip2 = B + shift2
        n_strip2_max = n_strip2 if sup is None else sup["B_max"] + shift2
        pt2 = self._plan_eval(n_strip2, 2 * Ht2, W, dtype, partial=True)
        for p in (ps2, pt2):
            self._eval_prepare(p["blocks"], True)
        img_t2, img_s2 = (Ht2 + 1) * Wp2, (h2s + 1) * Wp2
        stream2_base = n_strip2_max * img_t2                       # first row of the phase-0 stream image in cat2
        rows2 = stream2_base + 2 * img_s2 + W2 + 2
        d = 0 if sup is None else sup["d"]                          # this group's first window within the run (even)
        if sup is None:
            cat2 = pw.get("l2cat")
            if cat2 is None or cat2.numel() != rows2 * C2:
                cat2 = pw["l2cat"] = torch.zeros(rows2 * C2, device=self.device, dtype=dtype)
            stream_base = n_strip * (Ht + 1) * Wp
        else:
            cat2 = sup.get("cat2")
            stream_base = sup["stream_row0"]
        streams_ready = cat2 is not None and sup is not None
        if sup is not None and cat2 is None:
            cat2 = sup["cat2"] = self._sup_buffer("l2", rows2 * C2, dtype, (sup["S"], sup["B_max"], H, W))
            ps2 = self._plan_eval(2, 2 * h2s, W, dtype, partial=True, owner="l2")   # (a replaced buffer took the old run's plans with it)
            self._eval_prepare(ps2["blocks"], True)
        out_t2 = cat2[:(n_strip2 * img_t2 + W2 + 2) * C2]
        if n_strip2 < n_strip2_max:
            cat2[n_strip2 * img_t2 * C2:(n_strip2 * img_t2 + W2 + 2) * C2].zero_()
        out_s2 = cat2[stream2_base * C2:]
        stream_row0 = stream_base + d * Wp                          # row 0 of this group's window 0 in the level-1 stream
        stream2_row0 = stream2_base + (d // 2) * Wp2                # ... in the phase streams of level 2 (d even: phases keep their parity)

        def s2_launches(b, launch, launch_sc=None):
            """conv1 (-> slot 0) and the 1x1 shortcut (-> slot 1) of the down-sampling block b: one launch (launch_sc) or two."""
            if launch_sc is not None and self._s2_shortcut_rides(b):
                label = f"conv_f16_s2sc<{b.conv1.cin},{b.conv1.cout}>"
                t0 = self._mark(label)
                launch_sc(b)
                self._mark_end(label, t0)
                return
            for cs_, bn, slot, relu in ((b.conv1, b.bn1, 0, 1), (b.sc_conv, b.sc_bn, 1, 0)):
                label = f"conv_f16_s2<{cs_.cin},{cs_.cout},{cs_.taps}>"
                t0 = self._mark(label)
                launch(cs_, bn, slot, relu)
                self._mark_end(label, t0)

        def sc_args(b):
            return (_hip.ptr(b.conv1.wt_h), _hip.ptr(b.bn1.fold[0]), _hip.ptr(b.bn1.fold[1])), \
                   (_hip.ptr(b.sc_conv.wt_h), _hip.ptr(b.sc_bn.fold[0]), _hip.ptr(b.sc_bn.fold[1]))

        def rest_of_level(p, b, L, n_img, final_out):
            n_after = k3 - n1 - 1
            y = final_out if n_after == 0 else L[2]
            self._conv_eval_f16(b.conv2, b.bn2, L[0], L[1], y, n_img, 1)
            if n_after:
                self._eval_blocks(True, p, p["blocks"][n1 + 1:k3], y, n_img, final_out=final_out)

        # the two phase streams (once per run)
        Ls = None if streams_ready else ps2["lv"][(h2s, W2)]
        bs = ps2["blocks"][n1]
        for phase in (() if streams_ready else (0, 1)):
            src = ctypes.c_void_p(cat.data_ptr() + (stream_base + phase * Wp) * C1 * esize)

            def launch(cs_, bn, slot, relu, src=src, phase=phase):
                dst = ctypes.c_void_p(Ls[slot].data_ptr() + phase * img_s2 * C2 * esize)
                _hip.check(lib.lad_f16_conv_s2_fwd(src, _hip.ptr(cs_.wt_h), _hip.ptr(bn.fold[0]), _hip.ptr(bn.fold[1]), dst, 1, Hs_S, W,
                                                   cs_.cin, cs_.cout, cs_.taps, relu, st), "lad_f16_conv_s2_fwd " + cs_.name)

            def launch_sc(b, src=src, phase=phase):
                off = phase * img_s2 * C2 * esize
                c1, c2 = sc_args(b)
                _hip.check(lib.lad_f16_conv_s2_fwd_sc(src, *c1, ctypes.c_void_p(Ls[0].data_ptr() + off), *c2,
                                                      ctypes.c_void_p(Ls[1].data_ptr() + off), 1, Hs_S, W, b.conv1.cin, b.conv1.cout, 1, st),
                           "lad_f16_conv_s2_fwd_sc " + b.conv1.name)
            s2_launches(bs, launch, launch_sc)
        if not streams_ready:
            rest_of_level(ps2, bs, Ls, 2, out_s2)
        # the strips: the first and the last Ht2 rows of every window
        Lt = pt2["lv"][(Ht2, W2)]
        bt = pt2["blocks"][n1]

        def launch_t(cs_, bn, slot, relu):
            _hip.check(lib.lad_f16_conv_s2_fwd_mapped(_hip.ptr(cat), _hip.ptr(cs_.wt_h), _hip.ptr(bn.fold[0]), _hip.ptr(bn.fold[1]),
                                                      _hip.ptr(Lt[slot]), B, H, W, band, Ht, H - Ht, stream_row0, 1, 0, cat_rows, Ht2,
                                                      cs_.cin, cs_.cout, cs_.taps, relu, st), "lad_f16_conv_s2_fwd_mapped " + cs_.name)

        def launch_t_sc(b):
            c1, c2 = sc_args(b)
            if self.strip2_resident and (b.conv1.cin, b.conv1.cout) == (64, 32):
                # the strips' input rows resident in LDS as parity classes instead of gathered per lane (csrc/s2strip_f16.hip, round 6)
                rc = lib.lad_f16_conv_s2_strips_fwd(_hip.ptr(cat), *c1, _hip.ptr(Lt[0]), *c2, _hip.ptr(Lt[1]), B, H, W, band, Ht, H - Ht,
                                                    stream_row0, cat_rows, Ht2, st)
                if rc != _hip.LAD_NOT_COVERED:
                    _hip.check(rc, "lad_f16_conv_s2_strips_fwd " + b.conv1.name)
                    return
            _hip.check(lib.lad_f16_conv_s2_fwd_mapped_sc(_hip.ptr(cat), *c1, _hip.ptr(Lt[0]), *c2, _hip.ptr(Lt[1]), B, H, W, band, Ht, H - Ht,
                                                         stream_row0, 1, 0, cat_rows, Ht2, b.conv1.cin, b.conv1.cout, 1, st),
                       "lad_f16_conv_s2_fwd_mapped_sc " + b.conv1.name)
        s2_launches(bt, launch_t, launch_t_sc)
        rest_of_level(pt2, bt, Lt, n_strip2, out_t2)
        # level 3 onwards: per window -- in ONE launch where the tail kernel covers the geometry (csrc/tail_f16.hip, round 6)
        if self.tail_fused and self._tail_blocks_ok(blocks[k3:]):
            probs = self._probs_out if self._probs_out is not None else pw["probs"]
            label = "tail_f16"
            t0 = self._mark(label)
            rc = lib.lad_f16_tail_fwd(_hip.ptr(cat2), B, H2, W2, band2, Ht2, shift2, stream2_row0, 2, img_s2, rows2, self._tail_params(blocks[k3:]),
                                      self._head_params, pw["feat"], _hip.ptr(probs), st)
            if rc != _hip.LAD_NOT_COVERED:
                _hip.check(rc, "lad_f16_tail_fwd")
                self._mark_end(label, t0)
                return probs
        b3 = blocks[k3]
        L3 = pw["lv"][(b3.conv1.h_out, b3.conv1.w_out)]

        def launch_w(cs_, bn, slot, relu):
            _hip.check(lib.lad_f16_conv_s2_fwd_mapped(_hip.ptr(cat2), _hip.ptr(cs_.wt_h), _hip.ptr(bn.fold[0]), _hip.ptr(bn.fold[1]),
                                                      _hip.ptr(L3[slot]), B, H2, W2, band2, Ht2, shift2, stream2_row0, 2, img_s2, rows2, 0,
                                                      cs_.cin, cs_.cout, cs_.taps, relu, st), "lad_f16_conv_s2_fwd_mapped " + cs_.name)

        def launch_w_sc(b):
            c1, c2 = sc_args(b)
            _hip.check(lib.lad_f16_conv_s2_fwd_mapped_sc(_hip.ptr(cat2), *c1, _hip.ptr(L3[0]), *c2, _hip.ptr(L3[1]), B, H2, W2, band2, Ht2,
                                                         shift2, stream2_row0, 2, img_s2, rows2, 0, b.conv1.cin, b.conv1.cout, 1, st),
                       "lad_f16_conv_s2_fwd_mapped_sc " + b.conv1.name)
        s2_launches(b3, launch_w, launch_w_sc)
        self._conv_eval_f16(b3.conv2, b3.bn2, L3[0], L3[1], L3[2], B, 1)
        cur = self._eval_blocks(True, pw, blocks[k3 + 1:], L3[2], B)
        return self._eval_tail(True, pw, cur, B)

    def _stream_super_cap(self, F):
        """Windows per run of shared streams that the device's free memory allows (half of it): per frame a run keeps five
        64-channel tensors at level 1 (four rotating plan buffers + the stream) and five 32-channel ones per phase at level 2."""
        free, _ = torch.cuda.mem_get_info(self.device)
        free += torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)   # (torch's cache is reusable)
        held = sum(b.numel() * b.element_size() for b in self._sup_cache.values())                  # (a run of this size replaces it)
        per_frame = 2 * (6 * (F + 1) * self.stem_cout + 5 * ((F + 1) // 2 + 1) * 32)   # (+ the stem's output at level 1, kept for the strips)
        return max(0, int(0.5 * (free + held)) // per_frame)

    @staticmethod
    def _tail_blocks_ok(tail):
        """The layers lad_f16_tail_fwd runs: two (down-sampling block with a 1x1 shortcut, identity block) pairs, 32 -> 16 -> 16 channels."""
        if len(tail) != 4:
            return False
        for k, b in enumerate(tail):
            down = k % 2 == 0
            cin = 32 if k == 0 else 16
            if (b.conv1.cin, b.conv1.cout, b.conv1.taps, b.conv1.stride) != (cin, 16, 9, 2 if down else 1):
                return False
            if (b.conv2.cin, b.conv2.cout, b.conv2.taps, b.conv2.stride) != (16, 16, 9, 1):
                return False
            if down != (b.sc_conv is not None) or (down and (b.sc_conv.cin, b.sc_conv.cout, b.sc_conv.taps, b.sc_conv.stride) != (cin, 16, 1, 2)):
                return False
        return True

    def _tail_params(self, tail):
        """HOST array of the 30 device pointers lad_f16_tail_fwd takes: {fp16 weight image, folded scale, folded shift} per convolution
        (the tensors are created once per layer table by _fold_eval / _pack_f16 and refreshed in place: the pointers never move)."""
        key = id(tail[0])
        cached = self._tail_param_cache.get(key)
        ptrs = []
        for b in tail:
            for cs, bn in ((b.conv1, b.bn1),) + (((b.sc_conv, b.sc_bn),) if b.sc_conv is not None else ()) + ((b.conv2, b.bn2),):
                ptrs += [cs.wt_h.data_ptr(), bn.fold[0].data_ptr(), bn.fold[1].data_ptr()]
        if cached is None or cached[0] != ptrs:
            cached = self._tail_param_cache[key] = (ptrs, (_VP * len(ptrs))(*ptrs))
        return cached[1]

    def predict_windows(self, feats, n_frames=100, chunk=None, start=0, stop=None, out=None, precision="fp32", stream=True):
        """Probabilities of the stride-one-frame windows of a whole-file feature matrix (the loop of
        segment_laughter.py:90-101 over InferenceDataset, datasets.py:72-93): window i = feats[i:i+n_frames],
        zero-padded on the right at the end of the file.  feats: GPU float32 (T, F).  Windows [start, stop) only
        (rank sharding); returns a GPU float32 vector of stop-start probabilities.  precision "fp16" runs the
        convolutions on the 16-bit matrix cores with half activations (tolerance: tests/test_resnet_gpu.py).
        stream (default): the full-resolution layers are shared between the overlapping windows (_forward_eval_stream);
        False: every window goes through the whole model on its own, as the reference's loop does."""
        if precision not in ("fp32", "fp16"):
            raise ValueError("precision must be 'fp32' or 'fp16'")
        half = precision == "fp16"
        chunk = PREDICT_CHUNK[precision] if chunk is None else chunk
        self.ensure_flat()
        _hip.require_cuda(feats, "feats", torch.float32)
        if feats.dim() != 2:
            raise ValueError("feats must be (T, F)")
        T, F = feats.shape
        stop = T if stop is None else min(stop, T)
        n = max(0, stop - start)
        if out is None:
            out = torch.empty(n, device=feats.device, dtype=torch.float32)
        elif out.dim() != 1 or out.numel() < n:
            raise ValueError(f"out must be a vector of at least {n} elements (one per window of [start, stop))")
        flat = feats.view(-1)
        i = start
        direct = out.dtype == torch.float32 and out.is_contiguous() and out.device == feats.device
        # fp16: the streams of levels 1 and 2 once per RUN of groups (even group sizes: a group then starts at an even window of its run)
        use_runs = half and stream and self.stream_super and chunk % 2 == 0 and stop - start > chunk
        sup = None
        try:
            while i < stop:
                B = min(chunk, stop - i)
                dst = out[i - start:i - start + B]
                self._probs_out = dst if direct else None
                if use_runs and (sup is None or i >= sup["i0"] + sup["S"]):
                    S = min(stop - i, max(chunk, min(STREAM_SUPER_MAX, self._stream_super_cap(F)) // chunk * chunk))
                    sup = {"i0": i, "S": S, "B_max": min(chunk, S), "base": flat.data_ptr() + 4 * i * F, "frames_avail": T - i}
                if sup is not None:
                    sup["d"] = i - sup["i0"]
                if stream:
                    probs = self._forward_eval_stream(half, flat, B, n_frames, F, frames_avail=T - i, feat_offset_floats=i * F, sup=sup)
                else:
                    probs = self._forward_eval_any(half, flat, B, n_frames, F, 1, frames_avail=T - i, feat_offset_floats=i * F)
                if not direct:
                    dst.copy_(probs[:B])
                i += B
        finally:
            self._probs_out = None
        return out

    # ------------------------------------------------------------------------------------ backward
    def _bn_bwd(self, p, bn, dy, y, x, coef, dx, B, h, w, relu, mode=0, aux=None, sbn=None, xs=None, scoef=None, pre=False):
        """pre=True: the launch that produced dy already left this BatchNorm's per-tile sums in p["partials"]."""
        _hip.check(self.lib().lad_bn_bwd(
            _hip.ptr(dy), _hip.ptr(y), _hip.ptr(x), _hip.ptr(coef), _hip.ptr(bn.g), _hip.ptr(xs), _hip.ptr(scoef),
            _hip.ptr(sbn.g) if sbn is not None else None, _hip.ptr(dx), _hip.ptr(aux), _hip.ptr(bn.gg), _hip.ptr(bn.gb),
            _hip.ptr(sbn.gg) if sbn is not None else None, _hip.ptr(sbn.gb) if sbn is not None else None,
            _hip.ptr(p["bn_ws"]), _hip.ptr(p["bcoef"]), _hip.ptr(p["partials"]) if pre else None,
            int(self.lib().lad_conv_num_tiles(B, h, w)) if pre else 0, B, h, w, bn.c, relu, mode, self._st()), "lad_bn_bwd " + bn.name)

    # Weight gradients are off the critical path of backward (nothing needs them before the optimiser), so they CAN run
    # on a side stream next to the data-gradient chain (overlap_wgrad = True): two MFMA kernels sharing the CUs fill each
    # other's bubbles.  Off by default: every kernel then runs alone, and a per-kernel duration (the roofline figure of
    # bench.py, rocprofv3's averages) means what it says; with the overlap the step is faster but each co-scheduled
    # launch takes longer.  Ordering when on:
    #   side waits for main up to the launch point (its operands exist);
    #   main waits for the recorded side event before it OVERWRITES a gradient buffer a pending wgrad reads (_w);
    #   main joins side at the end of backward (the flat gradient is complete before all-reduce / clip / Adam).
    def _side_stream(self):
        if self._side is None or self._side.device != self.device:
            self._side = torch.cuda.Stream(self.device)
        return self._side

    def _on_side(self, launch, read_buffer, small=False):
        """Run launch(stream_handle) on the side stream; remember that it reads `read_buffer`.  small: a 16- / 32-channel layer's
        launch (overlap_wgrad_small: only these go to the side stream)."""
        if not (self.overlap_wgrad or (small and self._overlap_small_on())):
            launch(self._st())
            return
        side = self._side_stream()
        side.wait_stream(torch.cuda.current_stream(self.device))
        launch(ctypes.c_void_p(side.cuda_stream))
        ev = torch.cuda.Event()
        ev.record(side)
        self._side_readers[read_buffer.data_ptr()] = ev
        self._side_pending = True

    def _overlap_small_on(self):
        return self._cur_batch >= 256 if self.overlap_wgrad_small == "auto" else bool(self.overlap_wgrad_small)

    def _w(self, buf):
        """`buf` is about to be overwritten on the main stream: wait for a side-stream reader, if any."""
        ev = self._side_readers.pop(buf.data_ptr(), None)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
        return buf

    def _join_side(self):
        if self._side_pending:
            torch.cuda.current_stream(self.device).wait_stream(self._side)
            self._side_pending = False
            self._side_readers.clear()

    def _bnbwd_in_wgrad(self, cs):
        """Does this layer's weight gradient apply the BatchNorm backward of its own output (csrc/wgrad_mfma.hip, DOBN)?  The 64-channel
        f16 x 2 launches on the main stream only: the launch also writes the gradient its data-gradient launch reads."""
        return (self.fuse_bn_bwd_wgrad and not self.overlap_wgrad and self._use_b3(cs) and getattr(cs, "b3_wgrad", False)
                and self._h2(cs) and cs.cin == 64 and cs.cout == 64 and cs.stride == 1)

    def _wg_ws(self, p, cs):
        return p["wgrad_ws_of"][cs.name] if self._defer_on else p["wgrad_ws"]

    def _wgrad(self, p, cs, x, dout, B, h, w):
        lib = self.lib()
        if self._use_b3(cs) and getattr(cs, "b3_wgrad", False):
            # same split arithmetic as the forward / data-gradient launches of this layer (csrc/wgrad_mfma.hip)
            fn = lib.lad_conv_wgrad_h2 if (self._h2(cs) and cs.cin == 64) else lib.lad_conv_wgrad_b3c
            self._on_side(lambda st: _hip.check(fn(_hip.ptr(x), None, _hip.ptr(dout), _hip.ptr(self._wg_ws(p, cs)),
                                                                       _hip.ptr(cs.gw), _hip.ptr(cs.gb), B, h, w, cs.cin, st),
                                                "lad_conv_wgrad_b3c " + cs.name), dout, small=cs.cin <= 32)
            return
        self._on_side(lambda st: _hip.check(lib.lad_conv_wgrad(_hip.ptr(x), _hip.ptr(dout), _hip.ptr(self._wg_ws(p, cs)), _hip.ptr(cs.gw),
                                                               _hip.ptr(cs.gb), B, h, w, cs.cin, cs.cout, cs.taps, st),
                                            "lad_conv_wgrad " + cs.name), dout, small=cs.cin <= 32)

    def _dgrad(self, cs, dout, addend, dx, B, h, w, bnstat=None, partials=None):
        # data gradient = stride-1 convolution of dout with the flipped/transposed image: GEMM K = cout, N = cin.
        # bnstat = (x, y or None, coef) of the BatchNorm whose backward consumes dx: its first pass rides in the epilogue.
        fused_ok = bnstat is not None and self.fuse_bn_bwd and cs.taps == 9
        fused_b3 = (bnstat is not None and not fused_ok and self.fuse_bn_bwd_b3 and self._use_b3(cs) and bnstat[1] is None)
        label = self._split_label(cs, True) if (self._use_b3(cs) and not fused_ok) else f"conv_s1<{cs.cout},{cs.cin},{cs.taps}>"
        t0 = self._mark(label)
        if fused_b3:   # bn1 of a block: ReLU decisions recomputed from its input (csrc/conv_b3.hip, STAT epilogue)
            bx, _, bcoef = bnstat
            if self._h2(cs):
                self._conv_h2(cs, dout, None, cs.wt2_d, None, addend, None, dx, partials, bx, None, bcoef, B, h, w, "dgrad, bnstat")
            else:
                _hip.check(self.lib().lad_conv_b3c_dgrad_bnstat(_hip.ptr(dout), _hip.ptr(cs.wt3_d), _hip.ptr(addend), _hip.ptr(dx),
                                                                _hip.ptr(partials), _hip.ptr(bx), _hip.ptr(bcoef), B, h, w, cs.cin, self._st()),
                           "lad_conv_b3c_dgrad_bnstat " + cs.name)
            fused = True
        elif fused_ok:
            bx, by, bcoef = bnstat
            _hip.check(self.lib().lad_conv_fwd_bnstat(_hip.ptr(dout), _hip.ptr(cs.wt_d), _hip.ptr(addend), _hip.ptr(dx),
                                                      _hip.ptr(partials), _hip.ptr(bx), _hip.ptr(by), _hip.ptr(bcoef), B, h, w,
                                                      cs.cout, cs.cin, cs.taps, self._st()), "lad_conv_fwd_bnstat " + cs.name)
            fused = True
        else:
            self._dgrad_raw(cs, dout, addend, dx, B, h, w)
            fused = False
        self._mark_end(label, t0)
        return fused

    def _dgrad_raw(self, cs, dout, addend, dx, B, h, w):
        if self._use_b3(cs) and self._h2(cs):
            self._conv_h2(cs, dout, None, cs.wt2_d, None, addend, None, dx, None, None, None, None, B, h, w, "dgrad")
            return
        if self._use_b3(cs):
            _hip.check(self.lib().lad_conv_b3c_fwd_f32(_hip.ptr(dout), _hip.ptr(cs.wt3_d), None, _hip.ptr(addend), _hip.ptr(dx), None,
                                                       B, h, w, cs.cin, self._st()), "lad_conv_b3c_fwd_f32(dgrad) " + cs.name)
            return
        _hip.check(self.lib().lad_conv_fwd(_hip.ptr(dout), _hip.ptr(cs.wt_d), None, _hip.ptr(addend), _hip.ptr(dx), None, B, h, w,
                                           cs.cout, cs.cin, cs.taps, self._st()), "lad_conv_fwd(dgrad) " + cs.name)

    def backward(self, dprobs=None):
        """Gradient of the last train-mode forward into the flat gradient buffer (overwrites it).

        dprobs None: the loss is the mean BCE against the labels given to forward() (train.py:279-289);
        otherwise dprobs (B,) is dLoss/dprobs from autograd."""
        p = getattr(self, "_last_train_plan", None)
        live = {k: getattr(self, k) for k in self.KERNEL_OPTIONS}
        if p is not None and "options" in p:
            for k, v in p["options"].items():   # the forward pass's choices (see forward()); restored below
                setattr(self, k, v)
        try:
            self._backward(dprobs)
        except BaseException:
            if self._defer_on:   # leave the library's deferral switched off and its queue empty
                self._defer_on = False
                self.lib().lad_wgrad_defer_begin()
                self.lib().lad_wgrad_defer_flush(None)
            raise
        finally:
            for k, v in live.items():
                setattr(self, k, v)

    def _backward(self, dprobs):
        p = getattr(self, "_last_train_plan", None)
        if p is None or "saved" not in p:
            raise _hip.LadHipError("backward() without a preceding train-mode forward()")
        x, labels, m1, m2, B, H, W = p["saved"]
        if dprobs is None and labels is None:
            raise _hip.LadHipError("backward() needs dprobs or labels passed to forward()")
        lib, st = self.lib(), self._st()
        self._cur_batch = B
        blocks, acts = p["blocks"], p["acts"]
        last = blocks[-1].conv2
        # weight-gradient slab sums: one launch at the end instead of one per layer (not with the side stream: the flush
        # would have to follow launches on two streams)
        self._defer_on = self.defer_wgrad_sums and not self.overlap_wgrad
        if self._defer_on:
            _hip.check(lib.lad_wgrad_defer_begin(), "lad_wgrad_defer_begin")
        _hip.check(lib.lad_head_bwd(self._head_params, self._head_grads, _hip.ptr(p["pooled"]), _hip.ptr(p["h"]),
                                    _hip.ptr(p["hstats"]), _hip.ptr(p["probs"]), _hip.ptr(dprobs), B, p["feat"], _hip.ptr(m1),
                                    _hip.ptr(m2), _hip.ptr(labels), _hip.ptr(p["head_ws"]), _hip.ptr(p["dpooled"]), st),
                   "lad_head_bwd")
        g_out = p["g"][(last.h_out, last.w_out)]
        dy = g_out[0]
        _hip.check(lib.lad_pool_bwd(_hip.ptr(p["dpooled"]), _hip.ptr(dy), B, p["h4"], p["w4"], last.cout, st), "lad_pool_bwd")
        pre2 = False  # did the producer of `dy` already reduce for this block's bn2?
        pre2_tiles = 0  # ... into how many partials, if not one per 128-row tile (the stride-2 data gradient)
        for bi in range(len(blocks) - 1, -1, -1):
            b, a = blocks[bi], acts[bi]
            c1s, c2s = b.conv1, b.conv2
            ho, wo, co = c1s.h_out, c1s.w_out, c1s.cout
            hi, wi = c1s.h_in, c1s.w_in
            G = p["g"][(ho, wo)]
            free = [t for t in G if t is not dy]
            aux, da1 = (self._w(a["aux"]) if "aux" in a else free[0]), free[1]
            dc2, dc1 = self._w(a["dc2"]), self._w(a["dc1"])
            bits = a["ybits"] if a.get("bits_live") else None
            # the element-wise half of a 64-channel BatchNorm backward rides in the weight-gradient launch that consumes it
            # (lad_conv_wgrad_h2_bnbwd writes dc for the data-gradient launch): lad_bn_bwd* then only leaves the coefficients
            fuse2 = bits is not None and self._bnbwd_in_wgrad(c2s)
            fuse1 = c1s.stride == 1 and self._bnbwd_in_wgrad(c1s)
            if bits is not None:
                # dc2 only; the shortcut's share dy * [y > 0] is formed from dy and the bits in conv1's data gradient below
                _hip.check(lib.lad_bn_bwd_bits(_hip.ptr(dy), _hip.ptr(bits), _hip.ptr(a["c2"]), _hip.ptr(a["coef2"]), _hip.ptr(b.bn2.g),
                                               None if fuse2 else _hip.ptr(dc2), _hip.ptr(b.bn2.gg), _hip.ptr(b.bn2.gb), _hip.ptr(p["bn_ws"]),
                                               _hip.ptr(p["bcoef"]), _hip.ptr(p["partials"]) if pre2 else None,
                                               (pre2_tiles or int(lib.lad_conv_num_tiles(B, ho, wo))) if pre2 else 0, B, ho, wo, co, st),
                           "lad_bn_bwd_bits " + b.bn2.name)
            elif b.sc_conv is None:
                self._bn_bwd(p, b.bn2, dy, a["y"], a["c2"], a["coef2"], dc2, B, ho, wo, 1, mode=1, aux=aux, pre=pre2)
            else:
                self._bn_bwd(p, b.bn2, dy, a["y"], a["c2"], a["coef2"], dc2, B, ho, wo, 1, mode=2, aux=aux,
                             sbn=b.sc_bn, xs=a["cs"], scoef=a["coefs"])
            if fuse2:
                xin, xcoef = (a["c1"], a["coef1"]) if a.get("a1_virtual") else (a["a1"], None)
                _hip.check(lib.lad_conv_wgrad_h2_bnbwd(_hip.ptr(xin), _hip.ptr(xcoef), _hip.ptr(dy), _hip.ptr(a["c2"]), _hip.ptr(bits),
                                                       _hip.ptr(a["coef2"]), _hip.ptr(p["bcoef"]), _hip.ptr(dc2), _hip.ptr(self._wg_ws(p, c2s)),
                                                       _hip.ptr(c2s.gw), _hip.ptr(c2s.gb), B, ho, wo, c2s.cin, st),
                           "lad_conv_wgrad_h2_bnbwd " + c2s.name)
            elif a.get("a1_virtual"):
                wfn = lib.lad_conv_wgrad_h2 if (self._h2(c2s) and c2s.cin == 64) else lib.lad_conv_wgrad_b3c
                self._on_side(lambda sst, c2s=c2s, a=a, dc2=dc2, wfn=wfn: _hip.check(wfn(
                    _hip.ptr(a["c1"]), _hip.ptr(a["coef1"]), _hip.ptr(dc2), _hip.ptr(self._wg_ws(p, c2s)), _hip.ptr(c2s.gw), _hip.ptr(c2s.gb),
                    B, ho, wo, c2s.cin, sst), "lad_conv_wgrad_b3c(bnrelu) " + c2s.name), dc2, small=c2s.cin <= 32)
            else:
                self._wgrad(p, c2s, a["a1"], dc2, B, ho, wo)
            pre1 = self._dgrad(c2s, dc2, None, da1, B, ho, wo, bnstat=(a["c1"], None, a["coef1"]), partials=p["partials"])
            self._bn_bwd(p, b.bn1, da1, None, a["c1"], a["coef1"], None if fuse1 else dc1, B, ho, wo, 2, mode=0, pre=pre1)  # mask recomputed from c1
            pre2, pre2_tiles = False, 0
            if fuse1:
                _hip.check(lib.lad_conv_wgrad_h2_bnbwd(_hip.ptr(a["x"]), None, _hip.ptr(da1), _hip.ptr(a["c1"]), None, _hip.ptr(a["coef1"]),
                                                       _hip.ptr(p["bcoef"]), _hip.ptr(dc1), _hip.ptr(self._wg_ws(p, c1s)), _hip.ptr(c1s.gw),
                                                       _hip.ptr(c1s.gb), B, hi, wi, c1s.cin, st), "lad_conv_wgrad_h2_bnbwd " + c1s.name)
            if self.debug_capture is not None:
                self.debug_capture[b.name] = {"dy": dy.clone(), "dc2": dc2.clone(), "aux": aux.clone() if bits is None else None, "da1": da1.clone(),
                                              "dc1": dc1.clone()}
            if c1s.stride == 1:
                if not fuse1:
                    self._wgrad(p, c1s, a["x"], dc1, B, hi, wi)
                dx = dy  # dy is dead after the first bn_bwd; never aliases dc1 / aux
                # who consumes dx: the bn2 of the block below (identity shortcut only: its sums need y and c2), or the stem bn
                if bi == 0:
                    stat = None  # the stem's convolution output is not kept: its BatchNorm sums come from lad_stem_bn_bwd_sums
                elif blocks[bi - 1].sc_conv is None:
                    stat = (acts[bi - 1]["c2"], acts[bi - 1]["y"], acts[bi - 1]["coef2"])
                else:
                    stat = None
                if bits is not None:
                    label = self._split_label(c1s, True)
                    t0 = self._mark(label)
                    below = acts[bi - 1] if bi > 0 and acts[bi - 1].get("bits_live") else None
                    if below is not None and self.fuse_bn_bwd_b3 and self._h2(c1s):
                        self._conv_h2(c1s, dc1, None, c1s.wt2_d, None, dy, bits, dx, p["partials"], below["c2"], below["ybits"], below["coef2"],
                                      B, hi, wi, "dgrad, gated, bnstat")
                        pre2 = True
                    elif self._h2(c1s):
                        self._conv_h2(c1s, dc1, None, c1s.wt2_d, None, dy, bits, dx, None, None, None, None, B, hi, wi, "dgrad, gated")
                        pre2 = False
                    elif below is not None and self.fuse_bn_bwd_b3:   # + the sums of the block below's bn2 (its own sign bits)
                        _hip.check(lib.lad_conv_b3_dgrad_bnstat(_hip.ptr(dc1), _hip.ptr(c1s.wt3_d), _hip.ptr(dy), _hip.ptr(bits), _hip.ptr(dx),
                                                                _hip.ptr(p["partials"]), _hip.ptr(below["c2"]), _hip.ptr(below["ybits"]),
                                                                _hip.ptr(below["coef2"]), B, hi, wi, st),
                                   "lad_conv_b3_dgrad_bnstat " + c1s.name)
                        pre2 = True
                    else:
                        _hip.check(lib.lad_conv_b3_fwd_f32_gated(_hip.ptr(dc1), _hip.ptr(c1s.wt3_d), None, _hip.ptr(dy), _hip.ptr(bits),
                                                                 _hip.ptr(dx), None, B, hi, wi, st), "lad_conv_b3_fwd_f32_gated " + c1s.name)
                        pre2 = False
                    self._mark_end(label, t0)
                else:
                    pre2 = self._dgrad(c1s, dc1, aux, dx, B, hi, wi, bnstat=stat, partials=p["partials"])
                dy = dx
            else:
                # stride-2 block: gradients of conv1 and of the 1x1 shortcut at their true cost (csrc/conv_s2_bwd.hip)
                GI = p["g"][(hi, wi)]
                dx = GI[0]
                sc = b.sc_conv
                fuse_sc = self.fuse_s2_shortcut_wgrad
                if fuse_sc:   # conv1's and the shortcut's weight gradients in one launch (same input rows; csrc/conv_s2_bwd.hip)
                    self._on_side(lambda sst, c1s=c1s, sc=sc, dc1=dc1, aux=aux, xin=a["x"]: _hip.check(lib.lad_conv_s2_wgrad_fused(
                        _hip.ptr(xin), _hip.ptr(dc1), _hip.ptr(aux), _hip.ptr(self._wg_ws(p, c1s)), _hip.ptr(c1s.gw), _hip.ptr(c1s.gb),
                        _hip.ptr(sc.gw), B, hi, wi, c1s.cin, c1s.cout, sst), "lad_conv_s2_wgrad_fused " + c1s.name), dc1, small=c1s.cin <= 32)
                    if dc1.data_ptr() in self._side_readers:   # (it ran on the side stream) the launch reads aux as well
                        self._side_readers[aux.data_ptr()] = self._side_readers[dc1.data_ptr()]
                else:
                    self._on_side(lambda sst, c1s=c1s, dc1=dc1, xin=a["x"]: _hip.check(lib.lad_conv_s2_wgrad(
                        _hip.ptr(xin), _hip.ptr(dc1), _hip.ptr(self._wg_ws(p, c1s)), _hip.ptr(c1s.gw), _hip.ptr(c1s.gb), B, hi, wi,
                        c1s.cin, c1s.cout, 9, sst), "lad_conv_s2_wgrad " + c1s.name), dc1, small=c1s.cin <= 32)
                below = acts[bi - 1] if bi > 0 and acts[bi - 1].get("bits_live") else None
                if self._use_s2b3(b):
                    # both data gradients on the split-operand path, parity class by parity class (dgrad_s2b3_kernel); with the
                    # sums of the block below's bn2 in the epilogue when that block keeps sign bits
                    stat = self.fuse_bn_bwd_b3 and below is not None
                    n_part = int(lib.lad_conv_s2b3_dgrad_partials(B, hi, wi))
                    assert not stat or n_part * 2 * 64 <= p["partials"].numel()
                    _hip.check(lib.lad_conv_s2b3_dgrad(_hip.ptr(dc1), _hip.ptr(aux), _hip.ptr(c1s.wt3_s2d), _hip.ptr(dx),
                                                       _hip.ptr(p["partials"]) if stat else None, _hip.ptr(below["c2"]) if stat else None,
                                                       _hip.ptr(below["ybits"]) if stat else None, _hip.ptr(below["coef2"]) if stat else None,
                                                       B, hi, wi, st), "lad_conv_s2b3_dgrad " + c1s.name)
                    if stat:
                        pre2, pre2_tiles = True, n_part
                elif self.fuse_s2_shortcut and self.fuse_bn_bwd_b3 and below is not None and c1s.cin == 64 and c1s.cout == 32:
                    # ... and dx is final when it is written: the sums of the block below's bn2 ride in the epilogue
                    n_part = int(lib.lad_conv_s2_dgrad_partials(B, hi, wi))
                    assert n_part * 2 * 64 <= p["partials"].numel()
                    _hip.check(lib.lad_conv_s2_dgrad_fused_bnstat(_hip.ptr(dc1), _hip.ptr(c1s.wt_d), _hip.ptr(aux), _hip.ptr(sc.wt_d), _hip.ptr(dx),
                                                                  _hip.ptr(p["partials"]), _hip.ptr(below["c2"]), _hip.ptr(below["ybits"]),
                                                                  _hip.ptr(below["coef2"]), B, hi, wi, c1s.cin, c1s.cout, st),
                               "lad_conv_s2_dgrad_fused_bnstat " + c1s.name)
                    pre2, pre2_tiles = True, n_part
                elif self.fuse_s2_shortcut:   # both data gradients in one launch, dx written once
                    _hip.check(lib.lad_conv_s2_dgrad_fused(_hip.ptr(dc1), _hip.ptr(c1s.wt_d), _hip.ptr(aux), _hip.ptr(sc.wt_d), _hip.ptr(dx),
                                                           B, hi, wi, c1s.cin, c1s.cout, st), "lad_conv_s2_dgrad_fused " + c1s.name)
                else:
                    _hip.check(lib.lad_conv_s2_dgrad(_hip.ptr(dc1), _hip.ptr(c1s.wt_d), _hip.ptr(dx), B, hi, wi, c1s.cin, c1s.cout, 9, 0,
                                                     st), "lad_conv_s2_dgrad " + c1s.name)
                if not fuse_sc:
                    self._on_side(lambda sst, sc=sc, aux=aux, xin=a["x"]: _hip.check(lib.lad_conv_s2_wgrad(
                        _hip.ptr(xin), _hip.ptr(aux), _hip.ptr(self._wg_ws(p, sc)), _hip.ptr(sc.gw), None, B, hi, wi, sc.cin, sc.cout, 1,
                        sst), "lad_conv_s2_wgrad " + sc.name), aux, small=sc.cin <= 32)
                if not self.fuse_s2_shortcut:
                    _hip.check(lib.lad_conv_s2_dgrad(_hip.ptr(aux), _hip.ptr(sc.wt_d), _hip.ptr(dx), B, hi, wi, sc.cin, sc.cout, 1, 1, st),
                               "lad_conv_s2_dgrad " + sc.name)
                dy = dx
        # stem: bn1 + conv1 weight gradient.  The input needs no gradient and the convolution is recomputed from the features:
        # sums (x recomputed) -> lad_bn_bwd finalises them into dgamma / dbeta / bcoef (dx = None: nothing to apply) ->
        # the weight-gradient kernel applies the BatchNorm backward on the fly.  Neither x nor dz ever exist in HBM.
        if p.get("stem_mom_live"):
            # ONE pass over dy (the BatchNorm's sums and the centred tap products together), the rest from the forward's moments
            bn = self.stem_bn
            _hip.check(lib.lad_stem_bwd_onepass(_hip.ptr(x), _hip.ptr(self.stem_w), _hip.ptr(dy), _hip.ptr(p["stem_coef"]), _hip.ptr(bn.g),
                                                _hip.ptr(p["stem_mom"]), _hip.ptr(p["stem_bwd_ws"]), _hip.ptr(self.stem_gw), _hip.ptr(bn.gg),
                                                _hip.ptr(bn.gb), B, H, W, self.stem_cout, st), "lad_stem_bwd_onepass")
        else:
            groups = int(lib.lad_stem_bn_bwd_groups(B, H, W))
            _hip.check(lib.lad_stem_bn_bwd_sums(_hip.ptr(x), _hip.ptr(self.stem_w), _hip.ptr(dy), _hip.ptr(p["stem_coef"]),
                                                _hip.ptr(p["partials"]), B, H, W, self.stem_cout, st), "lad_stem_bn_bwd_sums")
            bn = self.stem_bn
            _hip.check(lib.lad_bn_bwd(_hip.ptr(dy), None, None, _hip.ptr(p["stem_coef"]), _hip.ptr(bn.g), None, None, None, None,
                                      None, _hip.ptr(bn.gg), _hip.ptr(bn.gb), None, None, _hip.ptr(p["bn_ws"]), _hip.ptr(p["bcoef"]),
                                      _hip.ptr(p["partials"]), groups, B, H, W, bn.c, 2, 0, st), "lad_bn_bwd " + bn.name)
            self._on_side(lambda sst: _hip.check(lib.lad_stem_wgrad_bn(_hip.ptr(x), _hip.ptr(dy), None, _hip.ptr(self.stem_w),
                                                                       _hip.ptr(p["stem_coef"]), _hip.ptr(p["bcoef"]),
                                                                       _hip.ptr(p["wgrad_ws"]), _hip.ptr(self.stem_gw), B, H, W,
                                                                       self.stem_cout, sst), "lad_stem_wgrad_bn"), dy)
        if self._defer_on:
            self._defer_on = False
            self._join_side()   # (the one launch that sums every layer's slabs follows the weight-gradient launches of both streams)
            _hip.check(lib.lad_wgrad_defer_flush(st), "lad_wgrad_defer_flush")
        self._join_side()
        self._grad_dirty = True

    def export_relu_masks(self):
        """ReLU decisions of the last train-mode forward, as CPU 0/1 tensors in the reference's (B,C,H,W) layout:
        {"stem", "block<k>.<j>.a1", "block<k>.<j>.y", "head"}.  For parity tests only (tests/test_resnet_gpu.py: with these
        decisions imposed on a CPU autograd run the two backward passes compute the same function)."""
        p = getattr(self, "_last_train_plan", None)
        if p is None or "saved" not in p:
            raise _hip.LadHipError("export_relu_masks() without a preceding train-mode forward()")
        x, labels, m1, m2, B, H, W = p["saved"]

        def unpack(buf, h, w, c):
            body = buf[:B * (h + 1) * (w + 1) * c].view(B, h + 1, w + 1, c)[:, 1:, 1:, :]
            return (body > 0).permute(0, 3, 1, 2).cpu()

        out = {"stem": unpack(p["stem_a"], H, W, self.stem_cout)}
        for b, a in zip(p["blocks"], p["acts"]):
            ho, wo, co = b.conv1.h_out, b.conv1.w_out, b.conv1.cout
            if a.get("a1_virtual"):
                # never stored: the sign of the kernels' fmaf(c1, scale, shift), taken from the same expression in double (the
                # product is exact there and the sum keeps its sign; a separate fp32 multiply + add does NOT always agree)
                c = a["c1"][:B * (ho + 1) * (wo + 1) * co].view(B, ho + 1, wo + 1, co)[:, 1:, 1:, :].double()
                sc, sh = a["coef1"][:co].double(), a["coef1"][co:2 * co].double()
                out[b.name + ".a1"] = (c * sc + sh > 0).permute(0, 3, 1, 2).cpu()
            else:
                out[b.name + ".a1"] = unpack(a["a1"], ho, wo, co)
            out[b.name + ".y"] = unpack(a["y"], ho, wo, co)
        # head: relu(dropout(bn3(h))) with batch statistics; h = linear1 output kept for the backward
        h = p["h"].view(B, 32).double()
        mean, var = h.mean(0), h.var(0, unbiased=False)
        u = (h - mean) / torch.sqrt(var + 1e-5) * self.head_bn3.g.double() + self.head_bn3.b.double()
        if m2 is not None:
            u = u * m2.double()
        out["head"] = (u > 0).cpu()
        return out

    # ------------------------------------------------------------------------------------ optimiser
    def reset_optimizer(self):
        """A fresh Adam, as run_epoch creates at train.py:336."""
        self.ensure_flat()
        self._exp_avg.zero_()
        self._exp_avg_sq.zero_()
        self._step_dev.zero_()
        self._step_count = 0

    def reset_dropout_rng(self):
        """Restart the sequence of dropout-mask draws of the fused step (masks are a function of torch's CUDA seed and the draw number; a
        change of the seed restarts it by itself, a torch.manual_seed with the SAME value cannot be told from no call)."""
        self.ensure_flat()
        self._rng_counter.zero_()

    def flat_grad(self):
        self.ensure_flat()
        return self._flat_g

    def flat_param(self):
        self.ensure_flat()
        return self._flat_p

    def accumulated_grad(self):
        """The accumulation buffer of the fused loop's gradient accumulation (zeros until accumulate_grad() is called)."""
        self.ensure_flat()
        if getattr(self, "_acc_g", None) is None:
            self._acc_g = torch.zeros_like(self._flat_g)
        return self._acc_g

    def accumulate_grad(self, scale):
        """accumulated_grad() += scale * flat_grad(): `(loss / gradient_accumulation_steps).backward()` of train.py:287-289
        (the engine's backward overwrites the flat gradient, so the running sum lives in a buffer of its own)."""
        acc = self.accumulated_grad()
        _hip.check(self.lib().lad_grad_accumulate(_hip.ptr(acc), _hip.ptr(self._flat_g), self._n_flat, float(scale), self._st()),
                   "lad_grad_accumulate")
        return acc

    def clip_and_step(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0, grad_scale=1.0, zero_grad=True, grad=None):
        """clip_grad_norm_(max_norm) + Adam.step() + zero_grad() (train.py:291-295) in two launches.
        grad_scale multiplies the gradient first (1/world_size after a sum all-reduce).  grad: the gradient buffer to step
        with (default: the flat gradient; the accumulation buffer under gradient accumulation)."""
        self.ensure_flat()
        lib, st = self.lib(), self._st()
        g = self._flat_g if grad is None else grad
        if g.numel() != self._n_flat or g.dtype != torch.float32 or not g.is_cuda:
            raise _hip.LadHipError("clip_and_step: grad must be a float32 GPU buffer of the flat parameter size")
        self._step_count += 1  # host mirror; the kernels use the device-side counter (hipGraph replays)
        _hip.check(lib.lad_grad_sumsq(_hip.ptr(g), self._n_flat, _hip.ptr(self._norm_partials), _hip.ptr(self._step_dev), st),
                   "lad_grad_sumsq")
        _hip.check(lib.lad_adam_step(_hip.ptr(self._flat_p), _hip.ptr(g), _hip.ptr(self._exp_avg),
                                     _hip.ptr(self._exp_avg_sq), self._n_flat, _hip.ptr(self._norm_partials), float(grad_scale),
                                     float(max_norm if max_norm is not None else 0.0), float(lr), float(betas[0]), float(betas[1]),
                                     float(eps), self._step_count, _hip.ptr(self._step_dev), 1 if zero_grad else 0,
                                     _hip.ptr(self._norm_out), st), "lad_adam_step")
        self.notify_weights_changed()
        if zero_grad:
            self._grad_dirty = False
        return self._norm_out

    def optimizer_state(self):
        """Tensors that make up the optimiser + parameter state (for snapshot / restore around graph warm-up)."""
        self.ensure_flat()
        return [self._flat_p, self._flat_g, self._exp_avg, self._exp_avg_sq, self._step_dev]

    def eval_metrics(self, probs, labels, out=None):
        """Counter vector (same layout as metrics()) of eval-mode probabilities against int32 labels, on the device."""
        _hip.require_cuda(probs, "probs", torch.float32)
        _hip.require_cuda(labels, "labels", torch.int32)
        if out is None:
            out = torch.zeros(8, device=probs.device)
        _hip.check(self.lib().lad_bce_metrics(_hip.ptr(probs), _hip.ptr(labels), probs.numel(), _hip.ptr(out), self._st()),
                   "lad_bce_metrics")
        return out

    def metrics(self):
        """float32[8] device tensor of the last train forward: mean BCE, #correct, #pred+, #true+, #target+, B."""
        return self._last_train_plan["metrics"]


def dropout_masks(B, feat, rate, device, generator=None):
    """Inverted-dropout masks for the two nn.Dropout calls of models.py:232,235 (torch RNG, tiny tensors)."""
    if rate <= 0.0:
        return None
    keep = 1.0 - rate
    m1 = torch.empty((B, feat), device=device).bernoulli_(keep, generator=generator).div_(keep)
    m2 = torch.empty((B, 32), device=device).bernoulli_(keep, generator=generator).div_(keep)
    return m1, m2


def metrics_from_counters(m):
    """(loss, accuracy, precision, recall) from the head's counter vector, as _calc_metrics (train.py:203-224)."""
    loss, corr, pp, tp, tt, n = [float(v) for v in m[:6]]
    acc = corr / n if n else float("nan")
    prec = 1.0 if pp == 0 else tp / pp
    rec = tp / tt if tt != 0 else float("nan")
    return loss, acc, prec, rec
