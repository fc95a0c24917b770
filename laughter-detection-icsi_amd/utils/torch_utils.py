"""Drop-in for the reference's utils/torch_utils.py (initialisation + checkpoint format).

Reference: utils/torch_utils.py:17-24 (count_parameters, init_weights: every parameter ~ N(0, 0.01), BatchNorm
gamma/beta included), :36-93 (save_checkpoint / load_checkpoint / make_state_dict: `last.pth.tar` + `best.pth.tar`
holding {'epoch','global_step','best_val_loss','state_dict','optim_dict'}), :98-101 (epoch_time).
Function names, arguments, file names and dictionary keys are the contract (checkpoints interchange with the reference
in both directions, tests/test_data_gpu.py); the bodies are this package's own.
"""
import os
import shutil

import torch

LAST, BEST = 'last.pth.tar', 'best.pth.tar'
CHECKPOINT_KEYS = ('epoch', 'global_step', 'best_val_loss', 'state_dict', 'optim_dict')


def count_parameters(model):
    n = 0
    for p in model.parameters():
        if p.requires_grad:
            n += p.numel()
    print(f'The model has {n:,} trainable parameters')
    return n


def init_weights(model):
    """Draw EVERY parameter of `model` (BatchNorm gamma/beta too) from N(0, 0.01), as the reference's initialiser does.
    Usable both directly and through `model.apply(init_weights)` (which re-draws once per submodule, as in the reference)."""
    with torch.no_grad():
        for p in model.parameters():
            p.normal_(mean=0.0, std=0.01)
    eng = getattr(model, "engine", None)
    if eng is not None and eng._flat_p is not None:
        eng.notify_weights_changed()  # packed MFMA images / BatchNorm folds derive from the parameters


def save_checkpoint(state, is_best, checkpoint):
    """Write `state` to <checkpoint>/last.pth.tar; when `is_best`, also keep a copy as best.pth.tar."""
    if not os.path.isdir(checkpoint):
        print(f"creating checkpoint directory {checkpoint}")
        os.makedirs(checkpoint, exist_ok=True)
    last = os.path.join(checkpoint, LAST)
    torch.save(state, last)
    if is_best:
        shutil.copyfile(last, os.path.join(checkpoint, BEST))


def load_checkpoint(checkpoint, model, optimizer=None, map_location=None):
    """Restore model (and optimiser, if both sides have one) from a checkpoint file; returns the loaded dictionary.
    Training resumes at the step AFTER the stored one (the reference stores the step it has just finished)."""
    if not os.path.isfile(checkpoint):
        raise FileNotFoundError(f"no checkpoint at {checkpoint}")
    ckpt = torch.load(checkpoint, map_location=map_location, weights_only=False)
    model.load_state_dict(ckpt['state_dict'])
    optim_state = ckpt.get('optim_dict')
    if optimizer is not None and optim_state is not None:
        optimizer.load_state_dict(optim_state)
    for attr in ('epoch', 'best_val_loss'):
        if attr in ckpt:
            setattr(model, attr, ckpt[attr])
    if 'global_step' in ckpt:
        model.global_step = ckpt['global_step'] + 1
    print(f"restored {checkpoint}: epoch {getattr(model, 'epoch', '?')}, resuming at step {getattr(model, 'global_step', '?')}")
    return ckpt


def make_state_dict(model, optimizer=None, epoch=None, global_step=None, best_val_loss=None):
    optim_state = None if optimizer is None else optimizer.state_dict()
    values = (epoch, global_step, best_val_loss, model.state_dict(), optim_state)
    return dict(zip(CHECKPOINT_KEYS, values))


def epoch_time(start_time, end_time):
    """Elapsed wall time as whole (minutes, seconds)."""
    mins, secs = divmod(int(end_time - start_time), 60)
    return mins, secs
