#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: 16 kHz PCM -> HIP fbank -> ResNetBigger fwd/bwd -> clip + Adam.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 512]

A step = one pass of the hot path over one batch of `--batch` synthetic 1 s clips per GPU that are already
resident in HBM (BASELINE.json configs[2]; for N > 1 configs[3]: data-parallel, one RCCL all-reduce of the flat
gradient per step, weak scaling).  Rank 0 prints ONE JSON line (contract in the task statement) carrying
  roofline     the dominant kernel (conv_s1<64,64,9>: the 64->64 3x3 convolutions of block1, forward and data
               gradient): algorithmic FLOPs per launch / mean launch duration from HIP events recorded on the launch
               stream inside the timed region, against the dense fp32-matrix MFMA peak;
  cpu_baseline the CPU oracle (oracle/, a port) timed on this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "laughter-detection-icsi_amd")
for p in (os.path.join(PKG, "utils"), PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X dense fp32-matrix peak (MI355X_MICROARCH.md, chip-level parameters)
DOMINANT = "conv_s1<64,64,9>"
DOMINANT_FLOP_PER_SEG = 2.0 * 100 * 44 * 64 * 64 * 9   # one 64->64 3x3 conv over a 100x44 map


def cpu_baseline(n_batches, batch=32):
    """The oracle (numpy fbank + torch-CPU functional ResNet step) on this host's cores: segments/s."""
    import numpy as np
    from oracle import fbank_oracle as fo, recipe, resnet_oracle as ro
    # the cores this process may actually run on (a GPU box hands a 1-GPU job a share of the host, not all of it)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16))
    torch.set_num_threads(threads)
    sd = ro.to_torch_state(recipe.make_state(101))
    adam, step = None, 0
    clips = recipe.make_clips(1234, batch)
    labels = torch.from_numpy(recipe.make_labels(4321, batch))
    # warm-up (thread pools, allocator)
    feats = torch.from_numpy(fo.fbank_batch(clips[:4], num_filters=44, dtype=np.float32))[:, None]
    ro.train_step(sd, feats, labels[:4])
    t0 = time.perf_counter()
    for _ in range(n_batches):
        feats = torch.from_numpy(fo.fbank_batch(clips, num_filters=44, dtype=np.float32))[:, None]
        r = ro.train_step(sd, feats, labels, adam_state=adam, step=step)
        sd, adam, step = r["new_sd"], r["adam_state"], r["step"]
    dt = time.perf_counter() - t0
    return {"value": round(n_batches * batch / dt, 2), "unit": "segments/s", "cores": threads, "kind": "port",
            "sample": f"{n_batches} batches of {batch} synthetic 1 s clips: numpy fbank + torch-CPU fp32 "
                      f"fwd/BCE/bwd/clip/Adam ({dt:.1f} s)"}


def side_workload(args):
    """BASELINE configs[1] and configs[4]: their own JSON line (not the driver's metric)."""
    import contextlib
    import io

    import config
    import parallel
    import synth
    from utils import get_feat_extractor
    rank, world, local = parallel.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ex = get_feat_extractor(config.FEAT["num_samples"], config.FEAT["num_filters"])
    if args.workload == "fbank":
        B = 1024
        pcm = synth.make_clips(B, seed=1234 + rank, device=dev)
        out = torch.empty((B, 100, 44), device=dev)
        for _ in range(args.warmup):
            ex.extract_batch(pcm, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            ex.extract_batch(pcm, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.steps
        gbs = B * 81600 / (ms * 1e-3) / 1e9
        print(json.dumps({"metric": "fbank segments/sec (HIP STFT->mel->log, batch 1024)", "value": round(B / (ms * 1e-3), 1),
                          "unit": "segments/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
                          "dtype": "f32", "data": "synthetic", "config": {"workload": "BASELINE configs[1]"},
                          "roofline": {"bound": "hbm", "kernel": "fbank_kernel", "achieved": round(gbs, 1), "peak": 8000.0,
                                       "unit": "GB/s", "frac": round(gbs / 8000.0, 4), "traffic": None,
                                       "bytes_per_segment": 81600}}), flush=True)
        return
    # infer
    cfg = config.MODEL_MAP["resnet_base"]
    with contextlib.redirect_stdout(io.StringIO()):
        model = cfg["model"](dropout_rate=0.0, linear_layer_size=cfg["linear_layer_size"], filter_sizes=cfg["filter_sizes"])
    model.set_device(dev)
    g = torch.Generator().manual_seed(9876)
    with torch.no_grad():  # non-degenerate random weights and running statistics (an init_weights model outputs a constant)
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
    model.eval()
    seconds = int(args.minutes * 60)
    pcm = synth.make_clips(seconds, seed=9876, device=dev).view(-1)  # one channel, generated second by second
    import time as _t
    import laugh_segmenter
    feats = ex.extract_long(pcm)
    T = feats.shape[0]
    sh = parallel.shard_indices(T, rank, world)
    prec = args.precision
    model.engine.predict_windows(feats, start=sh.start, stop=min(sh.stop, sh.start + 4096), precision=prec)  # warm-up
    torch.cuda.synchronize()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    t0 = _t.perf_counter()
    feats = ex.extract_long(pcm)
    local_p = model.engine.predict_windows(feats, start=sh.start, stop=sh.stop, precision=prec)
    probs = parallel.gather_probs(local_p, T, rank, world)
    torch.cuda.synchronize()
    t1 = _t.perf_counter()
    inst = laugh_segmenter.get_laughter_instances(probs.cpu().numpy(), [0.5], [0.2], 100.0)
    t2 = _t.perf_counter()
    if rank == 0:
        print(json.dumps({"metric": "sliding-window inference real-time factor (one %g min 16 kHz channel)" % args.minutes,
                          "value": round((t1 - t0) / seconds, 6), "unit": "s of compute per s of audio", "higher_is_better": False,
                          "n_gpus": world, "dtype": "f16" if prec == "fp16" else "f32", "data": "synthetic",
                          "config": {"workload": "BASELINE configs[4] (%s MFMA convolutions)" % prec, "windows": T,
                                     "windows_per_s": round(T / (t1 - t0), 1), "gpu_seconds": round(t1 - t0, 3),
                                     "segmenter_seconds": round(t2 - t1, 3), "instances": len(inst[(0.5, 0.2)])}}), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="segments per GPU per step")
    ap.add_argument("--dropout", type=float, default=0.5, help="train.py default")
    ap.add_argument("--cpu-batches", type=int, default=30, help="oracle batches of 32 for the CPU baseline (0 = skip)")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--fuse-bn-bwd", action="store_true",
                    help="train: BatchNorm-backward sums computed in the data-gradient epilogues (engine.fuse_bn_bwd; off by "
                         "default: it lengthens the dominant kernel's launches)")
    ap.add_argument("--overlap-wgrad", action="store_true",
                    help="weight gradients on a side stream (faster step; per-kernel durations then include co-scheduling)")
    ap.add_argument("--workload", default="train", choices=["train", "fbank", "infer"],
                    help="train: BASELINE configs[2]/[3] (the driver's metric).  fbank: configs[1] (HIP fbank only, batch 1024). "
                         "infer: configs[4] (sliding-window inference over one 60 min channel, real-time factor)")
    ap.add_argument("--minutes", type=float, default=60.0, help="infer: length of the synthetic channel")
    ap.add_argument("--precision", default="fp16", choices=["fp32", "fp16"], help="infer: matrix-core precision")
    args = ap.parse_args()
    if args.workload != "train":
        return side_workload(args)

    import parallel
    rank, world, local = parallel.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP hot path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import config
    import synth
    from engine import metrics_from_counters
    from utils import get_feat_extractor

    cfg = config.MODEL_MAP["resnet_base"]
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        model = cfg["model"](dropout_rate=args.dropout, linear_layer_size=cfg["linear_layer_size"],
                             filter_sizes=cfg["filter_sizes"])
    torch.manual_seed(1234)
    model.set_device(dev)
    import torch.nn as nn
    for name, param in model.named_parameters():  # utils/torch_utils.py:22-24 init_weights
        nn.init.normal_(param.data, mean=0, std=0.01)
    parallel.broadcast_parameters(model)
    model.train()
    model.engine.reset_optimizer()
    model.engine.overlap_wgrad = bool(args.overlap_wgrad)
    model.engine.fuse_bn_bwd = bool(args.fuse_bn_bwd)
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
        model.engine.kernel_events = {DOMINANT: []}
    distributed = torch.distributed.is_initialized()
    if distributed:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        met = step()
    torch.cuda.synchronize()
    if distributed:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    events = model.engine.kernel_events
    model.engine.kernel_events = None
    loss = metrics_from_counters(met.cpu().numpy())[0]
    if rank == 0:
        roof = None
        if events and events[DOMINANT]:
            ms = [a.elapsed_time(b) for a, b in events[DOMINANT]]
            avg_ms = sum(ms) / len(ms)
            flop = DOMINANT_FLOP_PER_SEG * B
            ach = flop / (avg_ms * 1e-3) / 1e12
            traffic = None  # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, guide's correction)
            pmc = os.path.join(ROOT, "profiles", "r01_conv_s1_pmc.json")
            if os.path.exists(pmc) and B == 512:
                traffic = json.load(open(pmc))["hbm_bytes_per_launch"]
            roof = {"bound": "mfma", "kernel": DOMINANT, "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                    "avg_launch_ms": round(avg_ms, 4), "launches_timed": len(ms),
                    "flop_per_launch": flop}
        cpu = None
        if args.cpu_batches > 0 and world == 1:
            cpu = cpu_baseline(args.cpu_batches)
        seg_s = world * B * args.steps / dt
        out = {
            "metric": "1 s@16 kHz segments/sec (featurize+ResNet fwd/bwd)",
            "value": round(seg_s, 1), "unit": "segments/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: end-to-end featurize (HIP fbank 44 mel) + ResNetBigger "
                                   "resnet_base fwd/bwd + clip + Adam, random labels, dropout %.1f" % args.dropout,
                       "segments_per_gpu_per_step": B, "global_batch": B * world,
                       "parallelism": f"dp{world}", "final_loss": round(loss, 5), "overlap_wgrad": bool(args.overlap_wgrad), "fuse_bn_bwd": bool(args.fuse_bn_bwd)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if distributed:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
