"""Drop-in for the reference's models.py on the MI355X: same classes, constructor arguments, attributes and
state_dict; the arithmetic runs in liblad_hip.so (hand-written HIP for gfx950) instead of torch kernels.

Reference surface mirrored here
    models.py:82-115   ResidualBlock(in_channels, out_channels, stride=1)
    models.py:181-244  ResNetBigger(num_classes=1, dropout_rate=0.5, linear_layer_size=192, filter_sizes=[64,32,16,16])
                       .forward(x: (B,1,T,F)) -> (B,1) sigmoid probabilities, .set_device(device),
                       attributes global_step / epoch / best_val_loss (models.py:211-213)
The torch.nn layers below are *parameter containers only* (they give the reference's state_dict keys, incl.
`num_batches_tracked`, so `best.pth.tar` / `last.pth.tar` checkpoints interchange); none of their forward methods
is ever called.  `forward` hands the batch to engine.ResNetEngine; under autograd it returns a tensor whose
backward runs the HIP backward and fills `param.grad` (views of one flat gradient buffer), so the reference's
`loss.backward(); clip_grad_norm_(...); optimizer.step(); model.zero_grad()` sequence (train.py:289-295) works
unchanged.  `train_step` is the sync-free fused equivalent of train.py:261-297 used by the fast loop.

There is no CPU fallback: calling the model with CPU tensors raises.
"""
import numpy as np
import torch
import torch.nn as nn

import _hip
from engine import ResNetEngine, dropout_masks, metrics_from_counters


class ResidualBlock(nn.Module):
    """Parameter container with the reference block's layout (models.py:82-108)."""

    def __init__(self, in_channels, out_channels, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=(3, 3), stride=stride, padding=1, bias=True)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=(3, 3), stride=1, padding=1, bias=True)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_channels != out_channels:
            self.shortcut = nn.Sequential(
                nn.Conv2d(in_channels, out_channels, kernel_size=(1, 1), stride=stride, bias=False),
                nn.BatchNorm2d(out_channels))

    def forward(self, x):
        raise _hip.LadHipError("ResidualBlock is executed by ResNetBigger's HIP engine, not on its own")


class _HipResNetFunction(torch.autograd.Function):
    """Connects the engine to autograd: forward = HIP forward, backward = HIP backward into param.grad."""

    @staticmethod
    def forward(ctx, x, hook, model, masks):
        ctx.model = model
        probs = model._engine.forward(x, train=True, drop_masks=masks)
        return probs.clone().view(-1, 1)

    @staticmethod
    def backward(ctx, dprobs):
        ctx.model._backward_from_autograd(dprobs.contiguous().view(-1).to(torch.float32))
        return None, None, None, None


class ResNetBigger(nn.Module):
    def __init__(self, num_classes=1, dropout_rate=0.5, linear_layer_size=192, filter_sizes=[64, 32, 16, 16]):
        super().__init__()
        print(f"training with dropout={dropout_rate}")
        if num_classes != 1:
            raise ValueError("the HIP head implements the reference's single-logit classifier (num_classes=1)")
        filter_sizes = list(filter_sizes)
        if len(filter_sizes) != 4:
            raise ValueError("filter_sizes must list the 4 stage widths")
        self.conv1 = nn.Conv2d(1, 64, kernel_size=(3, 3), stride=1, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.linear_layer_size = linear_layer_size
        self.filter_sizes = filter_sizes
        self.block1 = self._create_block(64, filter_sizes[0], stride=1)
        self.block2 = self._create_block(filter_sizes[0], filter_sizes[1], stride=2)
        self.block3 = self._create_block(filter_sizes[1], filter_sizes[2], stride=2)
        self.block4 = self._create_block(filter_sizes[2], filter_sizes[3], stride=2)
        self.bn2 = nn.BatchNorm1d(linear_layer_size)
        self.bn3 = nn.BatchNorm1d(32)
        self.linear1 = nn.Linear(linear_layer_size, 32)
        self.linear2 = nn.Linear(32, num_classes)
        self.dropout = nn.Dropout(dropout_rate)
        self.global_step = 0
        self.epoch = 0
        self.best_val_loss = np.inf
        # not modules / parameters: kept out of state_dict
        object.__setattr__(self, "_engine", ResNetEngine(self))
        object.__setattr__(self, "_hook", None)

    def _create_block(self, in_channels, out_channels, stride):
        return nn.Sequential(ResidualBlock(in_channels, out_channels, stride),
                             ResidualBlock(out_channels, out_channels, 1))

    # ---------------------------------------------------------------------------------------------- plumbing
    def set_device(self, device):
        for b in [self.block1, self.block2, self.block3, self.block4]:
            b.to(device)
        self.to(device)

    @property
    def engine(self):
        return self._engine

    def _bn_modules(self):
        return [m for m in self.modules() if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d))]

    def _bump_num_batches_tracked(self):
        self._engine.bump_num_batches_tracked()

    def zero_grad(self, set_to_none=True):
        self._engine._grad_dirty = False
        super().zero_grad(set_to_none=set_to_none)

    def _backward_from_autograd(self, dprobs):
        eng = self._engine
        prev = None
        if eng._grad_dirty and any(p.grad is not None for p in self.parameters()):
            prev = eng.flat_grad().clone()  # gradient accumulation across backward() calls
        eng.backward(dprobs)
        if prev is not None:
            eng.flat_grad().add_(prev)
        eng.attach_grads()

    # ---------------------------------------------------------------------------------------------- forward
    def forward(self, x):
        """x: (B,1,T,F) float GPU tensor -> (B,1) probabilities (models.py:222-239)."""
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise _hip.LadHipError("ResNetBigger.forward needs a GPU tensor (the HIP path has no CPU fallback)")
        x = x.to(torch.float32).contiguous()
        eng = self._engine
        eng.ensure_flat()
        if self.training:
            B = x.shape[0]
            masks = dropout_masks(B, self.linear_layer_size, float(self.dropout.p), x.device)
            self._bump_num_batches_tracked()
            if torch.is_grad_enabled():
                if self._hook is None or self._hook.device != x.device:
                    object.__setattr__(self, "_hook", torch.zeros(1, device=x.device, requires_grad=True))
                return _HipResNetFunction.apply(x, self._hook, self, masks)
            return eng.forward(x, train=True, drop_masks=masks).clone().view(-1, 1)
        return eng.forward(x, train=False).clone().view(-1, 1)

    # ---------------------------------------------------------------------------------------------- fused step
    def train_step(self, x, labels, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0, grad_reduce=None,
                   grad_scale=1.0, drop_masks="auto", grad_accum=1):
        """One optimisation step of train.py:261-297 without a host synchronisation.

        x: (B,1,T,F) / (B,T,F) float32 GPU features; labels: (B,) int32 GPU.  Runs forward (batch-stat BN, dropout),
        mean-BCE + metric counters, backward, optional `grad_reduce(flat_grad)` (data-parallel all-reduce),
        clip_grad_norm_(max_norm), Adam, zero_grad.  Returns the device tensor of counters
        (engine.metrics_from_counters turns it into loss / accuracy / precision / recall).

        grad_accum = gradient_accumulation_steps of train.py:287-295: every batch adds grad(loss) / grad_accum to a running
        sum; the optimiser steps (all-reduce, clip, Adam, zero) on the batches where `global_step % grad_accum == 0` --
        the reference's condition, evaluated before global_step is incremented: batch 0, then every grad_accum-th."""
        if not self.training:
            raise _hip.LadHipError("train_step() on a model in eval mode: call model.train() first")
        eng = self._engine
        eng.ensure_flat()
        x = x.to(torch.float32).contiguous()
        if labels.dtype != torch.int32:
            labels = labels.to(torch.int32)
        if isinstance(drop_masks, str) and drop_masks == "auto":
            drop_masks = "rng"       # masks drawn inside the head's launch, which also advances num_batches_tracked (engine.forward)
        else:
            self._bump_num_batches_tracked()
        eng.forward(x, train=True, labels=labels.contiguous(), drop_masks=drop_masks)
        eng.backward(None)
        grad_accum = int(grad_accum)
        if grad_accum <= 1:
            if grad_reduce is not None:
                grad_reduce(eng.flat_grad())
            eng.clip_and_step(lr=lr, betas=betas, eps=eps, max_norm=max_norm, grad_scale=grad_scale, zero_grad=True)
        else:
            acc = eng.accumulate_grad(1.0 / grad_accum)
            if self.global_step % grad_accum == 0:
                if grad_reduce is not None:
                    grad_reduce(acc)
                eng.clip_and_step(lr=lr, betas=betas, eps=eps, max_norm=max_norm, grad_scale=grad_scale, zero_grad=True, grad=acc)
        self.global_step += 1
        return eng.metrics()

    def make_graphed_train_step(self, batch_size, n_frames=100, n_filters=44, extractor=None, n_samples=16000, **opt):
        """Capture one train_step (optionally preceded by the fbank launch) for a fixed batch size in a hipGraph and
        return `step(inputs, labels) -> counters`.  The step is ~170 short launches; below a few dozen segments per
        batch (the reference trains with 32, load_data.py:32) it is launch-bound and a graph replay removes the
        per-launch host cost.  inputs: (B,1,T,F)/(B,T,F) features, or (B, n_samples) PCM when `extractor` is given.
        Single-process only (a captured RCCL all-reduce is not exercised here); dropout masks come from torch's
        graph-safe generator, the Adam step number from the device-side counter."""
        if not self.training:
            raise _hip.LadHipError("make_graphed_train_step() on a model in eval mode")
        if opt.get("grad_reduce") is not None:
            raise _hip.LadHipError("the graphed step is single-process: pass no grad_reduce")
        eng = self._engine
        eng.ensure_flat()
        dev = eng.device
        B = int(batch_size)
        x_static = torch.zeros((B, n_samples) if extractor is not None else (B, 1, n_frames, n_filters), device=dev)
        feats = torch.zeros((B, n_frames, n_filters), device=dev) if extractor is not None else None
        labels_static = torch.zeros(B, device=dev, dtype=torch.int32)

        def body():
            if extractor is not None:
                extractor.extract_batch(x_static, out=feats)
                return self.train_step(feats, labels_static, **opt)
            return self.train_step(x_static, labels_static, **opt)

        # warm-up outside the graph (plans, kernel attributes, RNG registration) on throw-away state
        state = eng.optimizer_state() + list(self.buffers())
        saved = [t.clone() for t in state]
        gs, host_steps = self.global_step, eng._step_count
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            labels_static.bernoulli_(0.5)
            for _ in range(2):
                body()
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            met = body()
        with torch.no_grad():
            for t, s0 in zip(state, saved):
                t.copy_(s0)
        self.global_step, eng._step_count = gs, host_steps
        eng.notify_weights_changed()

        def step(inputs, labels):
            x_static.copy_(inputs.reshape(x_static.shape))
            labels_static.copy_(labels)
            graph.replay()
            self.global_step += 1
            eng._step_count += 1
            # the replay moved the flat parameters and the running statistics on the device; the host-side cache tags
            # of the packed weight images / BatchNorm folds / fp16 packs must move with them (engine._state_tag)
            eng.notify_weights_changed()
            eng._train_forwards += 1
            return met

        step.graph = graph
        return step

    @torch.no_grad()
    def predict(self, x):
        """Eval-mode probabilities (B,) for a batch of windows; result is a plan-owned buffer (copy to keep)."""
        return self._engine.forward(x.to(torch.float32).contiguous(), train=False)


__all__ = ["ResidualBlock", "ResNetBigger", "metrics_from_counters"]
