"""Drop-in for the reference's utils/torch_utils.py (initialisation + checkpoint format).

Reference: utils/torch_utils.py:17-24 (count_parameters, init_weights: every parameter ~ N(0, 0.01), BatchNorm
gamma/beta included), :36-93 (save_checkpoint / load_checkpoint / make_state_dict: `last.pth.tar` + `best.pth.tar`
holding {'epoch','global_step','best_val_loss','state_dict','optim_dict'}), :98-101 (epoch_time).
The drop-in ResNetBigger keeps the reference's state_dict keys, so checkpoints interchange in both directions.
"""
import os
import shutil

import torch
import torch.nn as nn


def count_parameters(model):
    counts = sum(p.numel() for p in model.parameters() if p.requires_grad)
    print(f'The model has {counts:,} trainable parameters')


def init_weights(model):
    for name, param in model.named_parameters():
        nn.init.normal_(param.data, mean=0, std=0.01)


def save_checkpoint(state, is_best, checkpoint):
    filepath = os.path.join(checkpoint, 'last.pth.tar')
    if not os.path.exists(checkpoint):
        print("Checkpoint Directory does not exist! Making directory {}".format(checkpoint))
        os.makedirs(checkpoint, exist_ok=True)
    torch.save(state, filepath)
    if is_best:
        shutil.copyfile(filepath, os.path.join(checkpoint, 'best.pth.tar'))


def load_checkpoint(checkpoint, model, optimizer=None, map_location=None):
    if not os.path.exists(checkpoint):
        raise FileNotFoundError("File doesn't exist {}".format(checkpoint))
    print("Loading checkpoint at:", checkpoint)
    checkpoint = torch.load(checkpoint, map_location=map_location, weights_only=False)
    model.load_state_dict(checkpoint['state_dict'])
    if optimizer and checkpoint.get('optim_dict') is not None:
        optimizer.load_state_dict(checkpoint['optim_dict'])
    if 'epoch' in checkpoint:
        model.epoch = checkpoint['epoch']
    if 'global_step' in checkpoint:
        model.global_step = checkpoint['global_step'] + 1
        print("Loading checkpoint at step: ", model.global_step)
    if 'best_val_loss' in checkpoint:
        model.best_val_loss = checkpoint['best_val_loss']
    return checkpoint


def make_state_dict(model, optimizer=None, epoch=None, global_step=None, best_val_loss=None):
    return {'epoch': epoch, 'global_step': global_step, 'best_val_loss': best_val_loss,
            'state_dict': model.state_dict(),
            'optim_dict': optimizer.state_dict() if optimizer is not None else None}


def epoch_time(start_time, end_time):
    elapsed_time = end_time - start_time
    elapsed_mins = int(elapsed_time / 60)
    elapsed_secs = int(elapsed_time - (elapsed_mins * 60))
    return elapsed_mins, elapsed_secs
