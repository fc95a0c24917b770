#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: 16 kHz PCM -> HIP fbank -> ResNetBigger fwd/bwd -> clip + Adam.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 512]

A step = one pass of the hot path over one batch of `--batch` synthetic 1 s clips per GPU that are already
resident in HBM (BASELINE.json configs[2]; for N > 1 configs[3]: data-parallel, one RCCL all-reduce of the flat
gradient per step, weak scaling).  Rank 0 prints ONE JSON line (contract in the task statement) carrying
  roofline     the dominant kernel (conv_b3<64,64,9>: the eight 64->64 3x3 convolution launches of block1, forward and
               data gradient, on the bf16 matrix cores with three-way split operands): algorithmic FLOPs per launch /
               mean launch duration from HIP events recorded on the launch stream inside the timed region, against the
               roofline of that arithmetic (dense bf16 MFMA peak / 6 MFMAs per fp32-equivalent product); with --no-b3 the
               exact-f32 MFMA kernel conv_s1<64,64,9> against the dense fp32-matrix peak;
  cpu_baseline the CPU oracle (oracle/, a port) timed on this host by the protocol of SURVEY.md section 8(d) /
               BASELINE.md section 3 (C1: 256 clips at batch 32, eval and train, all cores and one thread, median of 3);
  side         (N = 1 only) BASELINE configs[1] (HIP fbank, 1024 clips) and configs[4] (fp16 sliding-window inference
               over a 60 min channel) as sub-records with their own rooflines: < 2 s of GPU time.

Timing.  After W warm-up steps the run times three blocks of exactly K steps each, every block bracketed by barrier +
synchronize on both sides and max-reduced over the ranks; `ms_per_step` and `value` are the MEDIAN block, the three
block times and their spread are in the line (a 0.3 s region is otherwise a single sample; boxes differ by 2-3 %).

Ranks.  With `--gpus N > 1` the N ranks come from an outer launcher (`python -m torch.distributed.run --nproc-per-node N
... bench.py --gpus N`, which exports RANK / WORLD_SIZE) or, when no launcher environment is present, from this script
itself: the parent starts N child processes BEFORE anything touches the GPU (parallel.spawn_ranks), relays rank 0's
line and exits with the worst child's code.  A world size that differs from --gpus, or fewer visible devices than
--gpus, is an error: the line never reports n_gpus for a job that did not run on that many GPUs.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "laughter-detection-icsi_amd")
for p in (os.path.join(PKG, "utils"), PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3    # MI355X dense fp32-matrix peak (MI355X_MICROARCH.md, chip-level parameters)
FP16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X dense fp16/bf16-matrix peak (same table; NOT the 2:1-sparsity figure)
HBM_PEAK_GBS = 8000.0
DOMINANT = "conv_s1<64,64,9>"      # the 64 -> 64 3x3 convolutions of block1 on the exact-f32 MFMA (--no-b3)
DOMINANT_B3 = "conv_b3<64,64,9>"   # the same launches on the bf16 matrix cores with three-way split operands (default)
DOMINANT_H2 = "conv_h2<64,64,9>"   # ... with two f16 planes per operand, three plane products (round 4, default)
DOMINANT_F16 = "conv_f16_s1<64,64,9>"
FUSED_BLOCK_F16 = "block_f16<64>"   # engine label of lad_f16_block_fwd: a residual block of the boundary strips in one launch (round 5)
DOMINANT_FLOP_PER_SEG = 2.0 * 100 * 44 * 64 * 64 * 9   # one 64->64 3x3 conv over a 100x44 map
FWD_FLOP_PER_SEG = 2.0 * 708330784                     # whole eval forward
FBANK_BYTES_PER_SEG = 81600                            # 64,000 B PCM read + 17,600 B features written
# per 10 ms frame: 256-point complex FFT 5 N log2 N = 10.2 k + real-FFT split 2.6 k + power 0.8 k + 44 triangular filters 1.0 k + window,
# pre-emphasis, mean 1.0 k = 15.6 kFLOP; 100 frames per segment (~1.5 MFLOP)
FBANK_FLOP_PER_SEG = 1.56e6


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """Cores this process may really use: its affinity mask, cut down to the cgroup CPU quota when there is one (a GPU box
    hands a one-GPU job a share of the host -- 16 CPUs -- while the affinity mask still lists every core of the host;
    as many threads as the mask lists would thrash inside that quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    source = "affinity"
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                q = int(float(quota) / period + 0.5)
                if 1 <= q < n:
                    n, source = q, "cgroup quota"
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n), source


def cpu_baseline(budget_s=25.0, batch=32, clips_c1=256):
    """SURVEY.md 8(d) / BASELINE.md section 3, on this host's cores with the oracle (a port: the reference itself cannot
    travel to the GPU box).  C1 = 256 synthetic clips in batches of 32 (load_data.py:32,53):
       (i) numpy fbank + eval forward     (ii) numpy fbank + forward / BCE / backward / clip / Adam
    each with every core this process may use AND with one thread, median of 3 repetitions.  The all-core legs run the
    whole of C1; the one-thread legs run as many batches of 32 as fit the time budget (stated in `sample`).
    `value` = leg (ii) on all cores: the same metric as the GPU line."""
    import numpy as np
    from oracle import fbank_oracle as fo, recipe, resnet_oracle as ro
    avail, core_source = usable_cores()
    sd0 = ro.to_torch_state(recipe.make_state(101))
    clips = recipe.make_clips(1234, clips_c1)
    labels = torch.from_numpy(recipe.make_labels(4321, clips_c1))

    def feats_of(lo):
        return torch.from_numpy(fo.fbank_batch(clips[lo:lo + batch], num_filters=44, dtype=np.float32))[:, None]

    def leg(train, n_batches):
        sd, adam, step = sd0, None, 0
        t0 = time.perf_counter()
        for b in range(n_batches):
            lo = (b * batch) % clips_c1
            x = feats_of(lo)
            if train:
                r = ro.train_step(sd, x, labels[lo:lo + batch], adam_state=adam, step=step)
                sd, adam, step = r["new_sd"], r["adam_state"], r["step"]
            else:
                with torch.no_grad():
                    ro.forward(sd, x, train=False)
        return n_batches * batch / (time.perf_counter() - t0)

    out = {}
    t_start = time.perf_counter()
    full = clips_c1 // batch
    plan = [("train_all_threads", True, avail, full), ("eval_all_threads", False, avail, full),
            ("eval_1_thread", False, 1, None), ("train_1_thread", True, 1, None)]
    notes = []
    for key, train, threads, n_batches in plan:
        name = key.split("_")[0]
        torch.set_num_threads(threads)
        leg(train, 1)  # warm-up: thread pool, allocator
        if n_batches is None:  # one-thread legs: size the sample from one timed batch so that 3 repetitions fit the budget
            one = batch / leg(train, 1)
            left = max(0.0, budget_s - (time.perf_counter() - t_start))
            share = left / (2.0 if name == "eval" else 1.0)
            n_batches = max(1, min(full, int(share / (3.0 * one))))
        reps = [leg(train, n_batches) for _ in range(3)]
        out[key] = round(statistics.median(reps), 2)
        notes.append(f"{name}/{threads}t: {n_batches}x{batch} clips")
    torch.set_num_threads(avail)
    dt = time.perf_counter() - t_start
    return {"value": out["train_all_threads"], "unit": "segments/s", "cores": avail, "kind": "port",
            "cpu_model": _cpu_model(), "host_cores": os.cpu_count(), "cores_from": core_source, "repetitions": 3,
            "statistic": "median",
            "train_all_threads": out["train_all_threads"], "eval_all_threads": out["eval_all_threads"],
            "train_1_thread": out["train_1_thread"], "eval_1_thread": out["eval_1_thread"],
            "sample": f"C1 ({clips_c1} synthetic 1 s clips, batch 32): numpy fbank + torch-CPU fp32 oracle; eval = forward only, "
                      "train = fwd/BCE/bwd/clip/Adam; per leg " + ", ".join(notes) + f"; {dt:.1f} s in all"}


# ------------------------------------------------------------------------------------------------ helpers
def _make_model(dropout, dev, degenerate_ok):
    """resnet_base on `dev`.  degenerate_ok: init_weights (utils/torch_utils.py:22-24, N(0, 0.01) everywhere), as
    train.py starts; otherwise non-degenerate random weights and running statistics (an init_weights model in eval
    mode outputs a constant)."""
    import contextlib
    import io

    import config
    import torch_utils
    cfg = config.MODEL_MAP["resnet_base"]
    with contextlib.redirect_stdout(io.StringIO()):
        model = cfg["model"](dropout_rate=dropout, linear_layer_size=cfg["linear_layer_size"], filter_sizes=cfg["filter_sizes"])
    model.set_device(dev)
    if degenerate_ok:
        torch.manual_seed(1234)
        model.apply(torch_utils.init_weights)
        return model
    g = torch.Generator().manual_seed(9876)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g) * (0.9 / (p[0].numel() ** 0.5)))
            elif name.endswith("weight"):
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
        for name, b in model.named_buffers():
            if name.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)
            elif name.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.2)
    model.engine.notify_weights_changed()
    return model


def _pmc_traffic(name):
    """HBM bytes per launch from a committed PMC summary (profiles/, tools/pmc_summary.py) + where it came from.  A summary carries
    the SHA-256 of the kernel's sources as they were when it was measured ("kernel_sources"): when one of them has changed since,
    or the summary predates that field, the value is not reported (None, "stale: ...") -- a counter must not outlive its kernel."""
    import hashlib
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, f"stale: profiles/{name} is missing"
    d = json.load(open(path))
    srcs = d.get("kernel_sources")
    if not srcs:
        return None, f"stale: profiles/{name} records no kernel source hash (measured before round 5)"
    for rel, want in srcs.items():
        try:
            have = hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest()
        except OSError:
            have = None
        if have != want:
            return None, f"stale: {rel} has changed since profiles/{name} was measured"
    return d.get("hbm_bytes_per_launch"), d.get("source", "profiles/" + name)


def fbank_record(ex, dev, steps, warmup, seed=1234):
    """BASELINE configs[1]: HIP STFT -> mel -> log over 1024 one-second clips resident in HBM."""
    import synth
    B = 1024
    pcm = synth.make_clips(B, seed=seed, device=dev)
    out = torch.empty((B, 100, 44), device=dev)
    for _ in range(warmup):
        ex.extract_batch(pcm, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        ex.extract_batch(pcm, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    gbs = B * FBANK_BYTES_PER_SEG / (ms * 1e-3) / 1e9
    traffic, src = _pmc_traffic("r05_fbank_pmc.json")
    # the same kernel over ONE 60 min channel (BASELINE configs[4]'s featurisation: 57.6 M samples -> (360000, 44)), where the
    # launch latency and the last partial round of workgroups no longer count
    channel = None
    if os.environ.get("LAD_BENCH_FBANK_CHANNEL", "1") != "0":   # (0: counter passes over the 1024-clip launches alone -- same kernel, same grid)
        long_pcm = synth.make_clips(3600, seed=9876, device=dev).view(-1)
        for _ in range(2):
            ex.extract_long(long_pcm)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            lf = ex.extract_long(long_pcm)
        e1.record()
        torch.cuda.synchronize()
        long_ms = e0.elapsed_time(e1) / 10
        long_bytes = long_pcm.numel() * 4 + lf.numel() * 4
        long_gbs = long_bytes / (long_ms * 1e-3) / 1e9
        del long_pcm, lf
        channel = {"ms": round(long_ms, 4), "algorithmic_bytes": int(long_bytes), "achieved": round(long_gbs, 1),
                   "frac": round(long_gbs / HBM_PEAK_GBS, 4),
                   "note": "one launch over 57.6 M samples -> (360000, 44): lad_fbank_forward_long"}
    return {"metric": "fbank segments/sec (HIP STFT->mel->log, batch 1024)", "value": round(B / (ms * 1e-3), 1),
            "unit": "segments/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "dtype": "f32", "data": "synthetic", "config": {"workload": "BASELINE configs[1]"},
            "roofline": {"bound": "hbm", "kernel": "fbank16_kernel", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": src,
                         "bytes_per_segment": FBANK_BYTES_PER_SEG,
                         # SURVEY 8(d) prices this kernel against HBM; at 18.4 FLOP/B it sits on the ridge of the f32 VECTOR roof (157.3 TF /
                         # 8 TB/s = 19.7 FLOP/B), so the same launch is priced there too -- and the counters say which one binds: neither
                         "bound_note": "arithmetic intensity 1.5 MFLOP / 81,600 B = 18.4 FLOP/B against a ridge of 157.3 TFLOP/s / 8 TB/s = 19.7: reaching "
                                       "the HBM peak would take 94 % of the f32 vector peak on butterflies that are not FMAs; PMC (profiles/r05_fbank_pmc.json): "
                                       "69 % of the wait cycles are LDS-issue stalls (SQ_WAIT_INST_LDS 21.4 M of 30.8 M), VALU 38 % busy -- the pass is a chain of "
                                       "~9 LDS round trips per four frames, not a stream",
                         "vector_side": {"flop_per_segment": FBANK_FLOP_PER_SEG, "achieved": round(B * FBANK_FLOP_PER_SEG / (ms * 1e-3) / 1e12, 2),
                                         "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s (f32 vector)",
                                         "frac": round(B * FBANK_FLOP_PER_SEG / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)},
                         "channel_60min": channel}}


def infer_record(ex, dev, minutes, precision, rank=0, world=1, emulate_world=None):
    """BASELINE configs[4]: featurise one synthetic channel, slide 1 s windows at one-frame stride through the eval model
    (fp16 or fp32 matrix cores), all-gather the probabilities, segment.  value = real-time factor."""
    import laugh_segmenter
    import parallel
    import synth
    model = _make_model(0.0, dev, degenerate_ok=False)
    model.eval()
    eng = model.engine
    seconds = int(minutes * 60)
    pcm = synth.make_clips(seconds, seed=9876, device=dev).view(-1)  # one channel, generated second by second
    feats = ex.extract_long(pcm)
    T = feats.shape[0]
    sh = parallel.shard_indices(T, rank, world)
    if emulate_world is not None:
        # what ONE rank of an `emulate_world`-GPU job does, timed on this GPU alone: the whole channel's featurisation (every rank
        # needs the features around its shard; the whole channel is 0.2 ms) + its shard of the windows.  No collective runs.
        sh = parallel.shard_indices(T, emulate_world - 1, emulate_world)   # (the last rank: its shard ends in the zero-padded windows)
    from engine import PREDICT_CHUNK
    # warm-up: one untimed pass over the shard (buffers are allocated on first use -- since round 5 also the run-long stream tensors,
    # 10 GB for this channel, which a warm-up of one group of windows would leave to the timed pass)
    n_local = sh.stop - sh.start
    eng.predict_windows(feats, start=sh.start, stop=sh.stop, precision=precision)
    torch.cuda.synchronize()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    label = DOMINANT_F16 if precision == "fp16" else DOMINANT
    # (an event pair costs the stream ~10 us: with the strips' blocks fused, only those two launches per group are timed)
    eng.kernel_events = {FUSED_BLOCK_F16: []} if precision == "fp16" and eng.strip_block_fused else {label: []}
    t0 = time.perf_counter()
    feats = ex.extract_long(pcm)
    local_p = eng.predict_windows(feats, start=sh.start, stop=sh.stop, precision=precision)
    if emulate_world is not None:
        torch.cuda.synchronize()
        eng.kernel_events = None
        shard_s = time.perf_counter() - t0
        return {"value": float("%.3g" % (shard_s / seconds)), "unit": "s of compute per s of audio", "gpus_emulated": emulate_world,
                "shard_windows": sh.stop - sh.start, "shard_seconds": round(shard_s, 4),
                "label": f"EMULATED on one GPU, no RCCL: the last of {emulate_world} ranks' work (whole-channel featurisation + its "
                         f"{sh.stop - sh.start} windows); a real run adds one all-gather of {T * 4} bytes of probabilities"}
    probs = parallel.gather_probs(local_p, T, rank, world)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    events, eng.kernel_events = eng.kernel_events, None
    inst = laugh_segmenter.get_laughter_instances(probs.cpu().numpy(), [0.5], [0.2], 100.0)
    t2 = time.perf_counter()
    gpu_s = t1 - t0
    peak = FP16_MFMA_PEAK_TFLOPS if precision == "fp16" else FP32_MFMA_PEAK_TFLOPS
    roof = None
    ms = [a.elapsed_time(b) for a, b in events.get(label, [])]
    ms_blk = [a.elapsed_time(b) for a, b in events.get(FUSED_BLOCK_F16, [])]
    if ms or ms_blk:
        chunk = PREDICT_CHUNK[precision]
        n_local = sh.stop - sh.start
        # launches of the dominant kernel come per chunk of windows, chunks in order.  Streaming path (engine default): the four
        # 64->64 convolutions of block1 run once over the chunk's frame stream (one image of w + 99 rows) and once over the
        # w + 90 boundary strips of 10 rows (one per frame offset: the top rows of one window and the bottom rows of another);
        # each launch is priced by the positions it really computes
        per_pos = DOMINANT_FLOP_PER_SEG / (100 * 44)
        flops, nbytes = [], []
        esize = 2 if precision == "fp16" else 4
        blk_flops, blk_bytes = [], []
        for i in range((n_local + chunk - 1) // chunk):
            w = min(chunk, n_local - i * chunk)
            fused = precision == "fp16" and eng.strip_block_fused and w >= 2 and w + 90 >= 256   # (engine._block_fits_lds)
            if fused:
                # the strips' two blocks are ONE launch each: both convolutions, one read and one write of the strip tensor
                rows = [(w + 99)] * 4
                blk_flops += [2 * per_pos * (w + 90) * 10 * 44] * 2
                blk_bytes += [2 * (w + 90) * 10 * 44 * 64 * esize] * 2
            else:
                rows = [(w + 99)] * 4 + [(w + 90) * 10] * 4 if w >= 2 else [100 * w] * 4
            flops += [per_pos * r * 44 for r in rows]
            # algorithmic bytes of a launch: its input and output tensor (44 x 64 elements per row), + the residual it adds in
            # the second convolution of a block (launch order: conv1, conv2, conv1, conv2)
            nbytes += [r * 44 * 64 * esize * (3 if k % 2 else 2) for k, r in enumerate(rows)]
        assert (len(flops) == len(ms) or not ms) and len(blk_flops) == len(ms_blk), (len(flops), len(ms), len(blk_flops), len(ms_blk))
        if ms_blk:
            # the dominant launches are the fused strip blocks: matrix-bound (the image stays in LDS between the convolutions:
            # 1,152 FLOP per byte moved against a ridge of 312), priced on the dense f16 peak
            big = [(f, b, t) for f, b, t in zip(blk_flops, blk_bytes, ms_blk) if f >= 0.5 * max(blk_flops)]
            t_big = sum(t for _, _, t in big) * 1e-3
            ach = sum(f for f, _, _ in big) / t_big / 1e12
            gbs = sum(b for _, b, _ in big) / t_big / 1e9
            traffic, src = _pmc_traffic("r06_block_f16_pmc.json")
            executed = sum(flops) + sum(blk_flops)
            roof = {"bound": "mfma", "kernel": FUSED_BLOCK_F16 + " (block_f16_strip_kernel: conv + BN + ReLU + conv + BN + residual + ReLU)",
                    "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "hbm_gbs_algorithmic": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_source": src, "avg_launch_ms": round(t_big * 1e3 / len(big), 4),
                    "launches_timed": len(ms_blk), "launches_priced": len(big), "flop_per_launch": big[0][0],
                    "bytes_per_launch": big[0][1],
                    "path": "streaming: levels 1 and 2 once over the frame stream + boundary strips (engine._forward_eval_stream)",
                    "executed_share_of_per_window_flops": round(executed / (4 * DOMINANT_FLOP_PER_SEG * n_local), 4),
                    "end_to_end_frac": round((n_local * FWD_FLOP_PER_SEG / gpu_s / 1e12) / peak, 4),
                    "end_to_end_note": "reference arithmetic per window (1.4167 GFLOP) / wall time / matrix peak: the streaming path executes less"}
            ms = []
    if ms:
        big = [(f, b, t) for f, b, t in zip(flops, nbytes, ms) if f >= 0.5 * max(flops)]   # the strip launches (90 % of the kernel's work)
        t_big = sum(t for _, _, t in big) * 1e-3
        ach = sum(f for f, _, _ in big) / t_big / 1e12
        gbs = sum(b for _, b, _ in big) / t_big / 1e9
        traffic, src = _pmc_traffic("r03_conv_f16_pmc.json" if precision == "fp16" else "r01_conv_s1_pmc.json")
        executed = sum(flops)
        common = {"kernel": label, "traffic": traffic, "traffic_source": src,
                  "avg_launch_ms": round(t_big * 1e3 / len(big), 4), "launches_timed": len(ms), "launches_priced": len(big),
                  "flop_per_launch": big[0][0], "bytes_per_launch": big[0][1],
                  "path": "streaming: levels 1 and 2 once over the frame stream + boundary strips (engine._forward_eval_stream)",
                  "executed_share_of_per_window_flops": round(executed / (4 * DOMINANT_FLOP_PER_SEG * n_local), 4),
                  "end_to_end_frac": round((n_local * FWD_FLOP_PER_SEG / gpu_s / 1e12) / peak, 4),
                  "end_to_end_note": "reference arithmetic per window (1.4167 GFLOP) / wall time / matrix peak: the streaming path executes less"}
        if precision == "fp16":
            # 64->64 3x3 in half precision: 73,728 FLOP per position over 256 B (384 B with the residual) = 288 (192) FLOP/B against
            # a ridge of 2500 / 8 = 312: the HBM roof is the lower one (and two restructurings of the kernel that cut its LDS
            # traffic / overlapped its phases changed nothing: profiles/r03_conv_f16_variants_ab.log)
            roof = dict({"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "mfma_tflops": round(ach, 1), "mfma_frac": round(ach / peak, 4)}, **common)
        else:
            roof = dict({"bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4)}, **common)
    return {"metric": "sliding-window inference real-time factor (one %g min 16 kHz channel)" % minutes,
            "value": float("%.3g" % (gpu_s / seconds)), "unit": "s of compute per s of audio", "higher_is_better": False,
            "n_gpus": world, "dtype": "f16" if precision == "fp16" else "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4] (%s MFMA convolutions)" % precision, "windows": T,
                       "windows_per_s": round(T / gpu_s, 1), "gpu_seconds": round(gpu_s, 3),
                       "segmenter_seconds": round(t2 - t1, 3), "instances": len(inst[(0.5, 0.2)]),
                       "tail_fused": bool(eng.tail_fused), "strip2_resident": bool(eng.strip2_resident)},
            "roofline": roof}


def _timed_leg(model, extractor, pcm, labels, args, reducer):
    """W warm-up + K timed steps of {fbank -> train_step} over the rotating batches pcm[i] / labels[i]: one timed block."""
    from engine import metrics_from_counters
    B, nb = args.batch, len(pcm)
    feats = torch.empty((B, 100, 44), device=pcm[0].device, dtype=torch.float32)

    def step(i):
        extractor.extract_batch(pcm[i % nb], out=feats)
        return model.train_step(feats, labels[i % nb], grad_reduce=reducer, grad_scale=reducer.scale)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        met = step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loss = metrics_from_counters(met.cpu().numpy())[0]
    return {"value": round(B * args.steps / dt, 1), "unit": "segments/s", "ms_per_step": round(1e3 * dt / args.steps, 3),
            "steps": args.steps, "final_loss": round(loss, 5)}


def realistic_leg(args, extractor, dev, reducer):
    """A second timed training leg on what a run in progress looks like, next to the headline's protocol state (init_weights
    = N(0, 0.01) everywhere as train.py starts, one batch of clips repeated): non-degenerate weights and running statistics
    (`_make_model(degenerate_ok=False)`) and FOUR distinct batches rotating.  The split-operand kernels' clock depends on the
    data they chew (profiles/README.md: 0.745 vs 0.85 ms per launch between the step's activations and N(0,1) operands), so
    this leg says how far the headline moves with the data.  Same step, same kernels, one timed block."""
    import synth
    model = _make_model(args.dropout, dev, degenerate_ok=False)
    model.train()
    model.engine.reset_optimizer()
    model.engine.bf16x3 = not args.no_b3
    model.engine.f16x2 = not args.no_h2
    B, nb = args.batch, 4
    pcm = [synth.make_clips(B, seed=777 + 31 * i, device=dev) for i in range(nb)]
    labels = [synth.make_labels(B, seed=555 + 17 * i, device=dev) for i in range(nb)]
    rec = _timed_leg(model, extractor, pcm, labels, args, reducer)
    rec.update({"state": "non-degenerate random weights (He-scaled) and running statistics, 4 distinct batches rotating",
                "headline_is": "the `value` of this line (the protocol's state: init_weights, one synthetic batch); this leg is the same "
                               "step on other data"})
    del model
    torch.cuda.empty_cache()
    return rec


def exact_legs(args, extractor, dev, reducer, pcm, labels):
    """The headline's step (same protocol state, same batch, one timed block of K steps) on the two arithmetics that carry every
    fp32 operand bit: three bf16 planes per operand (six plane products, `--no-h2`; exact operands, fp32 accumulation) and the
    exact-f32 MFMA (`--no-b3`: v_mfma_f32_32x32x2_f32 everywhere, bit-for-bit a fmaf chain).  The headline's default arithmetic
    (two f16 planes) carries 22-23 operand bits ("The arithmetic"): these legs say what the exact forms cost
    on the same box in the same run."""
    out = {}
    for name, flags, what in (
            ("bf16x3", {"f16x2": False, "f16x2_32": False},
             "64->64 / 32->32 convolutions on three bf16 planes per operand (6 MFMAs per product, exact operands): bench.py --no-h2"),
            ("f32", {"bf16x3": False},
             "every convolution on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32): bench.py --no-b3")):
        model = _make_model(args.dropout, dev, degenerate_ok=True)
        model.train()
        model.engine.reset_optimizer()
        for k, v in flags.items():
            setattr(model.engine, k, v)
        rec = _timed_leg(model, extractor, [pcm], [labels], args, reducer)
        rec["arithmetic"] = what
        out[name] = rec
        del model
        torch.cuda.empty_cache()
    return out


def side_workload(args):
    """BASELINE configs[1] and configs[4] on their own: one JSON line each (not the driver's metric)."""
    import config
    import parallel
    from utils import get_feat_extractor
    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP hot path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ex = get_feat_extractor(config.FEAT["num_samples"], config.FEAT["num_filters"])
    if args.workload == "fbank":
        rec = fbank_record(ex, dev, args.steps, args.warmup, seed=1234 + rank)
    else:
        rec = infer_record(ex, dev, args.minutes, args.precision, rank, world)
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=3, help="timed blocks of --steps steps each; the line reports the median block")
    ap.add_argument("--batch", type=int, default=512, help="segments per GPU per step")
    ap.add_argument("--dropout", type=float, default=0.5, help="train.py default")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="time budget of the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-clips", type=int, default=256, help="clips in the CPU baseline's C1 workload (256 = the protocol; tests shrink it)")
    ap.add_argument("--no-side", action="store_true", help="skip the configs[1] / configs[4] sub-records")
    ap.add_argument("--dump-params", default=None, help="rank 0 saves its flat parameter buffer here after the timed steps (tests: "
                                                        "the data-parallel result against a single-process emulation)")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-fuse-bnbwd-wgrad", action="store_true", help="train: the BatchNorm backward's element-wise pass as its own launch "
                    "(bn_bwd_apply_kernel) instead of inside the 64-channel weight-gradient launches (A/B)")
    ap.add_argument("--no-h2", action="store_true", help="train: the split-operand layers on three bf16 planes (rounds 2-3) instead of "
                    "two f16 planes (engine.f16x2 off)")
    ap.add_argument("--no-b3-32", action="store_true", help="train: the 32-channel convolutions on the exact-f32 MFMA (engine.bf16x3_32 off)")
    ap.add_argument("--no-relu-bits", action="store_true", help="train: residual ReLU masks re-read from y (engine.relu_bits off)")
    ap.add_argument("--no-fuse-sc", action="store_true",
                    help="train: the 1x1 shortcut's weight gradient in its own launch (engine.fuse_s2_shortcut_wgrad off)")
    ap.add_argument("--no-defer-sums", action="store_true",
                    help="train: one slab-sum launch per weight gradient instead of one per step (engine.defer_wgrad_sums off)")
    ap.add_argument("--no-virtual-a1", action="store_true",
                    help="train: write the activation between the two convolutions of a 64-channel block (engine.virtual_a1 off)")
    ap.add_argument("--no-fuse-b3", action="store_true",
                    help="train: BatchNorm-backward sums in their own passes, not in the bf16x3 data-gradient epilogues "
                         "(engine.fuse_bn_bwd_b3 off)")
    ap.add_argument("--no-b3", action="store_true",
                    help="64->64 convolutions on the exact-f32 MFMA instead of the bf16 matrix cores with three-way split "
                         "operands (engine.bf16x3; same fp32-level accuracy, 0.95 vs 1.31 ms per launch)")
    ap.add_argument("--fuse-bn-bwd", action="store_true",
                    help="train: BatchNorm-backward sums computed in the data-gradient epilogues (engine.fuse_bn_bwd; off by "
                         "default: it lengthens the dominant kernel's launches)")
    ap.add_argument("--overlap-small", action="store_true", help="train: the weight gradients of the 16- / 32-channel layers on the side "
                    "stream at ANY batch (the engine's default: from 256 segments per step on; -0.65 %% of the step at batch 512, "
                    "+3.5 %% at batch 32)")
    ap.add_argument("--no-overlap-small", action="store_true", help="train: ... on the main stream even at batch >= 256 (the A/B of the default)")
    ap.add_argument("--overlap-wgrad", action="store_true",
                    help="weight gradients on a side stream (faster step; per-kernel durations then include co-scheduling)")
    ap.add_argument("--workload", default="train", choices=["train", "fbank", "infer"],
                    help="train: BASELINE configs[2]/[3] (the driver's metric).  fbank: configs[1] (HIP fbank only, batch 1024). "
                         "infer: configs[4] (sliding-window inference over one 60 min channel, real-time factor)")
    ap.add_argument("--minutes", type=float, default=60.0, help="infer: length of the synthetic channel")
    ap.add_argument("--precision", default="fp16", choices=["fp32", "fp16"], help="infer: matrix-core precision")
    args = ap.parse_args()

    import parallel
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and not parallel.under_launcher():
        # parent launcher: this process has made no GPU call; it starts the ranks as fresh children and waits
        raise SystemExit(parallel.spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:], timeout=parallel.SPAWN_TIMEOUT_S))
    if args.workload != "train":
        return side_workload(args)

    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a figure under the wrong GPU count")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP hot path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import config
    import synth
    from engine import metrics_from_counters
    from utils import get_feat_extractor

    model = _make_model(args.dropout, dev, degenerate_ok=True)
    parallel.broadcast_parameters(model)
    model.train()
    model.engine.reset_optimizer()
    model.engine.overlap_wgrad = bool(args.overlap_wgrad)
    if args.overlap_small or args.no_overlap_small:   # (default: the engine's "auto" -- on from 256 segments per step)
        model.engine.overlap_wgrad_small = bool(args.overlap_small)
    model.engine.fuse_bn_bwd = bool(args.fuse_bn_bwd)
    model.engine.bf16x3 = not args.no_b3
    model.engine.relu_bits = not args.no_relu_bits
    model.engine.bf16x3_32 = not args.no_b3_32
    model.engine.f16x2 = not args.no_h2
    model.engine.fuse_bn_bwd_b3 = not args.no_fuse_b3
    model.engine.fuse_bn_bwd_wgrad = not args.no_fuse_bnbwd_wgrad
    model.engine.virtual_a1 = not args.no_virtual_a1
    model.engine.defer_wgrad_sums = not args.no_defer_sums
    model.engine.fuse_s2_shortcut_wgrad = not args.no_fuse_sc
    model.engine.fuse_s2_shortcut = not args.no_fuse_sc
    if os.environ.get("LAD_S2B3") == "0":   # (A/B knob) the 64 -> 32 stride-2 transition on round 2's f32 gather kernels
        model.engine.s2_b3 = False
    dominant = DOMINANT if args.no_b3 else (DOMINANT_B3 if args.no_h2 else DOMINANT_H2)
    extractor = get_feat_extractor(config.FEAT["num_samples"], config.FEAT["num_filters"])
    reducer = parallel.GradReducer()

    B = args.batch
    pcm = synth.make_clips(B, seed=1234 + rank, device=dev)
    labels = synth.make_labels(B, seed=4321 + rank, device=dev)
    feats = torch.empty((B, 100, 44), device=dev, dtype=torch.float32)

    def step():
        extractor.extract_batch(pcm, out=feats)
        return model.train_step(feats, labels, grad_reduce=reducer, grad_scale=reducer.scale)

    for _ in range(args.warmup):
        met = step()
    torch.cuda.synchronize()
    if not args.no_kernel_events:
        model.engine.kernel_events = {dominant: []}
        reducer.events = []
    calls0 = reducer.calls
    distributed = torch.distributed.is_initialized()
    def timed_block():
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            m = step()
        torch.cuda.synchronize()
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        d = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([d], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            d = float(t.item())
        return d, m

    blocks = []
    for _ in range(max(1, args.blocks)):
        d, met = timed_block()
        blocks.append(d)
    dt = statistics.median(blocks)

    events = model.engine.kernel_events
    model.engine.kernel_events = None
    if args.dump_params and rank == 0:
        torch.save(model.engine.flat_param().detach().cpu(), args.dump_params)
    loss = metrics_from_counters(met.cpu().numpy())[0]
    allreduce_ms = allreduce_ranks = None
    if reducer.events:
        mine = sum(a.elapsed_time(b) for a, b in reducer.events) / len(reducer.events)
        allreduce_ms = round(mine, 4)
        if distributed:   # min / median / max of the per-rank means
            every = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
            torch.distributed.all_gather(every, torch.tensor([mine], device=dev, dtype=torch.float64))
            v = sorted(float(t.item()) for t in every)
            allreduce_ranks = {"min": round(v[0], 4), "median": round(statistics.median(v), 4), "max": round(v[-1], 4)}
    reducer.events = None
    if rank == 0:
        roof = None
        if events and events[dominant]:
            ms = [a.elapsed_time(b) for a, b in events[dominant]]
            avg_ms = sum(ms) / len(ms)
            flop = DOMINANT_FLOP_PER_SEG * B
            ach = flop / (avg_ms * 1e-3) / 1e12
            traffic, src = (None, None)
            if args.no_b3:
                if B == 512:
                    # NOT measured inside this step: a rocprofv3 --pmc pass over the same kernel at the same shape in the
                    # micro-benchmark tools/bench_conv.py (FETCH_SIZE x2 + WRITE_SIZE, the guide's gfx950 correction)
                    traffic, src = _pmc_traffic("r01_conv_s1_pmc.json")
                    if traffic is not None:
                        src = "micro-benchmark tools/bench_conv.py under rocprofv3 --pmc (profiles/r01_conv_s1_pmc.json), not in-step"
                roof = {"bound": "mfma", "kernel": dominant, "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                        "traffic_source": src, "avg_launch_ms": round(avg_ms, 4), "launches_timed": len(ms),
                        "flop_per_launch": flop}
            elif args.no_h2:
                if B == 512:
                    # not measured by THIS run (counters need a profiler): the committed rocprofv3 --pmc passes over bench.py
                    # itself (same kernel, same shape, the step's own activations), or -- older -- over the micro-benchmark
                    traffic, src = _pmc_traffic("r03_conv_b3x_instep_pmc.json")
                    if traffic is not None:
                        src = ("in-step: rocprofv3 --pmc passes over `bench.py --steps 3` (profiles/r03_conv_b3x_instep_pmc.json; "
                               "mean over the step's eight launches), not this run")
                    else:
                        traffic, src = _pmc_traffic("r03_conv_b3x_pmc.json")
                        if traffic is not None:
                            src = "micro-benchmark tools/bench_conv.py convb3f under rocprofv3 --pmc (profiles/r03_conv_b3x_pmc.json), not in-step"
                # `achieved` = ALGORITHMIC FLOPs (2 * rows * 64 * 64 * 9) per launch, as for the f32 kernel.  The arithmetic is
                # fp32-equivalent on the bf16 pipe: SIX bf16 MFMAs per algorithmic product (three-way split operands), so the
                # roofline of this arithmetic is the dense bf16 peak / 6
                b3_peak = FP16_MFMA_PEAK_TFLOPS / 6.0
                roof = {"bound": "mfma", "kernel": dominant, "achieved": round(ach, 2), "peak": round(b3_peak, 1),
                        "unit": "TFLOP/s", "frac": round(ach / b3_peak, 4), "traffic": traffic,
                        "traffic_source": src, "avg_launch_ms": round(avg_ms, 4), "launches_timed": len(ms),
                        "flop_per_launch": flop,
                        "peak_derivation": "dense bf16 MFMA peak 2500 TFLOP/s / 6 MFMAs per fp32-equivalent product",
                        "arithmetic": "bf16 x 3 split operands, 6 MFMAs per product, f32 accumulate "
                                      "(fp32-equivalent: tests/test_resnet_gpu.py)",
                        "executed_matrix_tflops": round(6 * ach, 1), "executed_frac_of_bf16_peak": round(6 * ach / FP16_MFMA_PEAK_TFLOPS, 4),
                        "frac_of_fp32_matrix_peak": round(ach / FP32_MFMA_PEAK_TFLOPS, 4)}
            else:
                if B == 512:
                    traffic, src = _pmc_traffic("r06_conv_h2_instep_pmc.json")
                    if traffic is not None:
                        src = ("in-step: rocprofv3 --pmc passes over `bench.py --steps 3` (profiles/r06_conv_h2_instep_pmc.json; "
                               "mean over the step's eight launches), not this run")
                # `achieved` = ALGORITHMIC FLOPs (2 * rows * 64 * 64 * 9) per launch.  Two f16 planes per operand, THREE plane
                # products per algorithmic product (csrc/conv_h2.hip): the matrix roofline of this arithmetic is the dense f16
                # peak / 3.  The eight launches of a step also move 2-4 activation-sized tensors each (596 MB at batch 512:
                # input, output, + the addend / BatchNorm operands of the fused epilogues; mean 2.625): at this speed the
                # HBM side is as close as the matrix side, so both fractions are given
                h2_peak = FP16_MFMA_PEAK_TFLOPS / 3.0
                t_bytes = 4.0 * 64 * B * 101 * 45
                alg_bytes = 2.625 * t_bytes
                roof = {"bound": "mfma", "kernel": dominant, "achieved": round(ach, 2), "peak": round(h2_peak, 1),
                        "unit": "TFLOP/s", "frac": round(ach / h2_peak, 4), "traffic": traffic,
                        "traffic_source": src, "avg_launch_ms": round(avg_ms, 4), "launches_timed": len(ms),
                        "flop_per_launch": flop,
                        "peak_derivation": "dense f16 MFMA peak 2500 TFLOP/s / 3 MFMAs per fp32-equivalent product",
                        "arithmetic": "f16 x 2 split operands (block floating point per staged tile), 3 MFMAs per product, f32 "
                                      "accumulate; error vs float64 within 1.5x of the exact-f32 kernel's (tests/test_h2_gpu.py)",
                        "executed_matrix_tflops": round(3 * ach, 1), "executed_frac_of_f16_peak": round(3 * ach / FP16_MFMA_PEAK_TFLOPS, 4),
                        "frac_of_fp32_matrix_peak": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                        "hbm_side": {"algorithmic_bytes_per_launch_mean": int(alg_bytes),
                                     "achieved_GBps": round(alg_bytes / (avg_ms * 1e-3) / 1e9, 1), "peak_GBps": HBM_PEAK_GBS,
                                     "frac": round(alg_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     "launch_mix": "4 x (in + out), 2 x (in + out + BatchNorm input), 1 x (in + addend + out), "
                                                   "1 x (in + addend + out + BatchNorm input) per step"}}
        if roof is not None and model.engine._overlap_small_on():
            roof["co_scheduling"] = ("the weight gradients of the 16- / 32-channel layers run on a side stream (engine.overlap_wgrad_small, on from 256 "
                                     "segments per step): one or two of the step's eight launches of this kernel share the chip with them, and "
                                     "avg_launch_ms includes that (`--no-overlap-small`: 0.5524 ms, frac 0.3616, step +0.07 ms)")
        seg_s = world * B * args.steps / dt
        side = None
        if world == 1 and not args.no_side:
            side = {"fbank_1024": fbank_record(extractor, dev, 50, 10),
                    "infer_60min_fp16": infer_record(extractor, dev, 60.0, "fp16")}
            side["infer_60min_fp16"]["predicted_8gpu_rtf"] = infer_record(extractor, dev, 60.0, "fp16", emulate_world=8)
            side["train_realistic"] = realistic_leg(args, extractor, dev, reducer)
            if not (args.no_b3 or args.no_h2):
                side["train_exact"] = exact_legs(args, extractor, dev, reducer, pcm, labels)
        cpu = None
        if args.cpu_seconds > 0 and world == 1:
            cpu = cpu_baseline(args.cpu_seconds, clips_c1=args.cpu_clips)
        which = "configs[2]" if world == 1 else "configs[3]"
        out = {
            "metric": "1 s@16 kHz segments/sec (featurize+ResNet fwd/bwd)",
            "value": round(seg_s, 1), "unit": "segments/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "timed_blocks": {"statistic": "median", "ms_per_step": [round(1e3 * d / args.steps, 3) for d in blocks],
                             "spread_pct": round(100.0 * (max(blocks) - min(blocks)) / dt, 2)},
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32" if args.no_b3 else
                      "f32 (64->64 convolutions: bf16x3 split operands, f32 accumulate, fp32-equivalent)" if args.no_h2 else
                      "f32 (64->64 and 32->32 convolutions: f16x2 split operands with per-tile power-of-two scales, f32 accumulate; "
                      "error vs float64 within 1.5x of the exact-f32 kernels')"),
            "data": "synthetic",
            "rccl_ranks": torch.distributed.get_world_size() if distributed else 0,
            "allreduce_calls": (reducer.calls - calls0) // len(blocks),   # per timed block of K steps: one per step
            "allreduce_ms_per_step": allreduce_ms,
            "allreduce_ms_per_step_over_ranks": allreduce_ranks,
            "config": {"workload": f"BASELINE {which}: end-to-end featurize (HIP fbank 44 mel) + ResNetBigger "
                                   "resnet_base fwd/bwd + clip + Adam, random labels, dropout %.1f" % args.dropout
                                   + ("" if world == 1 else f"; data-parallel over {world} ranks, one RCCL all-reduce of the "
                                      "flat gradient (885 KB) per step"),
                       "segments_per_gpu_per_step": B, "global_batch": B * world,
                       "parallelism": f"dp{world}", "backend": reducer.backend, "final_loss": round(loss, 5),
                       "overlap_wgrad": bool(args.overlap_wgrad), "overlap_wgrad_small": bool(model.engine._overlap_small_on()),
                       "bn_bwd_in_wgrad": not args.no_fuse_bnbwd_wgrad and not args.no_h2 and not args.no_b3 and not args.overlap_wgrad,
                       "fuse_bn_bwd": bool(args.fuse_bn_bwd),
                       "bf16x3_convs": not args.no_b3, "f16x2_convs": not args.no_b3 and not args.no_h2, "relu_bits": not args.no_relu_bits and not args.no_b3,
                       "fuse_bn_bwd_b3": not args.no_fuse_b3 and not args.no_b3,
                       "virtual_a1": not args.no_virtual_a1 and not args.no_b3},
            "roofline": roof, "cpu_baseline": cpu, "side": side,
        }
        print(json.dumps(out), flush=True)
    if distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
