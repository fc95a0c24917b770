#!/usr/bin/env python3
"""Training loop on the MI355X: counterpart of the reference's train.py (argparse :68-117, `run_training_loop` :150-167,
`run_epoch` :170-415 with `_train_batch` :261-297 / `_eval_batch` :226-259 / `_eval_for_logging` :178-201, metrics CSV
:488-504, train_params.csv :314-322, checkpoints through utils/torch_utils.py).

Same flags, files and cadence; what changes is the step: `model.train_step` runs forward, BCE + metric counters,
backward, (all-reduce,) clip and Adam on the device without a host synchronisation -- the reference takes four `.item()`
round trips per step -- and the per-step counters are read back once per logging interval.  `--batch_size` is the real
batch size here (the reference's sampler pins 32 cuts regardless, load_data.py:32).  Launch with torchrun for
data-parallel training (one process per GPU, one RCCL all-reduce of the flat gradient per step).
Reference quirks kept on purpose: a fresh Adam per epoch (train.py:336), the learning-rate "schedule" is a no-op
(train.py:347-349 sets an unused attribute), checkpoint cadence == log cadence (train.py:159).
"""
import argparse
import csv
import os
import sys
import time

_PKG = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(_PKG, "utils"), _PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import config as config_mod  # noqa: E402
import load_data  # noqa: E402
import parallel  # noqa: E402
import torch_utils  # noqa: E402
from engine import metrics_from_counters  # noqa: E402

METRIC_COLS = ['batch_num', 'epoch', 'train_prec', 'train_rec', 'train_acc', 'train_loss', 'val_prec', 'val_rec', 'val_acc',
               'val_loss']


def batch_metrics(rows):
    """Per-batch (loss, acc, prec, rec) from counter rows, then the reference's aggregation (train.py:386-393):
    plain means, recall with nanmean."""
    m = np.array([metrics_from_counters(r) for r in rows], dtype=np.float64)
    with np.errstate(invalid="ignore"):
        rec = np.nanmean(m[:, 3]) if np.any(~np.isnan(m[:, 3])) else float("nan")
    return dict(loss=m[:, 0].mean(), acc=m[:, 1].mean(), prec=m[:, 2].mean(), rec=rec)


def eval_for_logging(model, val_iter_state, val_loader, n_batches, rank=0, world=1):
    """`n_batches` eval-mode batches from the validation loader (wrapping around), train.py:178-201.

    Data parallel: every rank walks the same validation ORDER (same seed) but fetches and evaluates only batches rank,
    rank + world, ... of the window -- the index batches of the others are drawn from the sampler and dropped, their
    features are never gathered; the counter rows of the others stay zero and one sum-all-reduce of the (n_batches, 8)
    matrix gives every rank all rows: the logged validation metrics are those of a single process over the same batches.
    `val_iter_state[0]` is an iterator over the loader's SAMPLER (index batches), or None to start one."""
    model.eval()
    rows = []
    it = val_iter_state[0]
    if it is None:
        it = iter(val_loader.sampler)
    with torch.no_grad():
        for i in range(max(1, n_batches)):
            try:
                idx = next(it)
            except StopIteration:
                it = iter(val_loader.sampler)
                idx = next(it)
            if i % world != rank:
                rows.append(None)
                continue
            batch = val_loader.dataset[idx]
            probs = model.predict(batch['inputs'])
            rows.append(model.engine.eval_metrics(probs, batch['is_laugh'].to(torch.int32)).clone())
    val_iter_state[0] = it
    model.train()
    mine = next(r for r in rows if r is not None) if any(r is not None for r in rows) else None
    zero = torch.zeros(8, device=mine.device) if mine is not None else torch.zeros(8, device=parallel.collective_device())
    mat = torch.stack([r if r is not None else zero for r in rows])
    return batch_metrics(parallel.sum_rows(mat).cpu().numpy())


def run_epoch(model, train_loader, val_loader, checkpoint_dir, log_frequency, batch_size, metrics_rows, reducer,
              rank=0, clip=1.0, verbose=True, max_steps=None, grad_accum=1):
    model.train()
    model.engine.reset_optimizer()  # optimizer = optim.Adam(model.parameters()) at the top of every epoch
    val_batches_per_log = 1
    if val_loader is not None:
        # batch_size is per rank: a step consumes batch_size * world segments, so an epoch has 1/world as many logging points
        # and each validates world times as many (per-rank-sized) batches -- the dev set is still walked once per training
        # epoch, and the window's batches are dealt to the ranks (eval_for_logging): per-rank validation cost is constant
        validations_per_epoch = train_loader.sampler.num_cuts / (batch_size * reducer.world * log_frequency)
        val_batches_per_log = int(val_loader.sampler.num_cuts / max(validations_per_epoch, 1e-9) / batch_size) or 1
    val_state = [None] if val_loader is not None else None
    hist = []
    epoch_loss_rows = []
    steps = 0
    for batch in train_loader:
        if batch['inputs'].shape[0] < 2:
            # BatchNorm needs more than one sample in train mode.  load_data's sampler never yields such a batch for the
            # train split (the decision is taken from the global sizes); a foreign loader may, and skipping it on one rank
            # only would leave the other ranks alone in their all-reduce
            if reducer.world > 1:
                raise RuntimeError("a training batch of fewer than 2 segments on one rank of a data-parallel job: use "
                                   "load_data.create_training_dataloader (rank-invariant batches)")
            continue
        if grad_accum > 1:
            met = model.train_step(batch['inputs'], batch['is_laugh'], max_norm=clip, grad_reduce=reducer, grad_scale=reducer.scale,
                                   grad_accum=grad_accum)
        else:
            met = model.train_step(batch['inputs'], batch['is_laugh'], max_norm=clip, grad_reduce=reducer, grad_scale=reducer.scale)
        hist.append(met.clone())
        steps += 1
        if log_frequency is not None and (model.global_step + 1) % log_frequency == 0:
            # the only device->host read of the interval; data parallel: step s of every rank is one GLOBAL batch, whose
            # counters are the sums over ranks (loss: mean weighted by the rank's batch size) -- parallel.reduce_counters
            train_m = batch_metrics(parallel.reduce_counters(torch.stack(hist)).cpu().numpy())
            epoch_loss_rows.append(train_m['loss'] * len(hist))
            hist = []
            is_best = False
            val_m = dict(loss=float('nan'), acc=float('nan'), prec=float('nan'), rec=float('nan'))
            if val_loader is not None:
                val_m = eval_for_logging(model, val_state, val_loader, val_batches_per_log, rank=rank, world=reducer.world)
                is_best = val_m['loss'] < model.best_val_loss
                if is_best:
                    model.best_val_loss = val_m['loss']
            metrics_rows.append([model.global_step, model.epoch, train_m['prec'], train_m['rec'], train_m['acc'], train_m['loss'],
                                 val_m['prec'], val_m['rec'], val_m['acc'], val_m['loss']])
            if verbose and rank == 0:
                print(f"\nLogging at step: {model.global_step}\nTrain metrics: {train_m}\nValidation metrics: {val_m}")
            if rank == 0:
                state = torch_utils.make_state_dict(model, None, model.epoch, model.global_step, model.best_val_loss)
                torch_utils.save_checkpoint(state, is_best=is_best, checkpoint=checkpoint_dir)
        if max_steps is not None and steps >= max_steps:
            break
    if hist:
        epoch_loss_rows.append(batch_metrics(parallel.reduce_counters(torch.stack(hist)).cpu().numpy())['loss'] * len(hist))
    model.epoch += 1
    return float(np.sum(epoch_loss_rows))  # the reference returns the SUM of batch losses (num_batches = +1, train.py:356)


def update_metrics_on_disk(metrics_file, rows):
    old = []
    if os.path.isfile(metrics_file):
        with open(metrics_file, newline='') as f:
            old = list(csv.reader(f))[1:]
    with open(metrics_file, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(METRIC_COLS)
        w.writerows(old + rows)


IGNORED_FLAGS = {
    'num_workers': "batches are gathered on the GPU from HBM-resident features (datasets.FeatureStore): there are no loader workers",
    'torch_device': "the HIP path has no CPU fallback; the device is cuda:<LOCAL_RANK>",
}


def warn_ignored_flags(args, parser, out=None):
    """Flags of the reference's train.py (:68-117) that are accepted for command-line compatibility and change nothing here:
    say so once instead of ignoring them silently."""
    out = out or sys.stderr
    for name, why in IGNORED_FLAGS.items():
        if getattr(args, name) != parser.get_default(name):
            out.write(f"train.py: --{name} {getattr(args, name)} has no effect ({why})\n")


def find_feats_manifests(args):
    """{split: manifest path} of the stored features to train from: --feats_manifest_dir, else the reference's place for
    them, <data_root>/<lhotse_dir>/cutsets (compute_features.py:105-111 writes, load_data.py:24-25 reads)."""
    d = args.feats_manifest_dir
    explicit = d is not None
    if d is None:
        d = os.path.join(args.data_root, args.lhotse_dir, 'cutsets')
    found = {}
    names = os.listdir(d) if os.path.isdir(d) else []
    for split in ('train', 'dev'):
        # a data-parallel compute_features run leaves {split}_feats.rank<r>.jsonl instead of {split}_feats.jsonl: the stem path
        # stands for them (load_data.read_feats_manifest gathers the rank files next to it)
        if f'{split}_feats.jsonl' in names or any(n.startswith(f'{split}_feats.rank') and n.endswith('.jsonl') for n in names):
            found[split] = os.path.join(d, f'{split}_feats.jsonl')
    if explicit and not found:
        raise SystemExit(f"--feats_manifest_dir {d}: no train_feats.jsonl / dev_feats.jsonl (or their .rank<r>.jsonl parts) there")
    return found


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--config', type=str, required=True)
    parser.add_argument('--checkpoint_dir', type=str, required=True)
    parser.add_argument('--data_root', type=str, required=True)
    parser.add_argument('--num_epochs', type=int, default=1)
    parser.add_argument('--lhotse_dir', type=str, default='lhotse')
    parser.add_argument('--data_dfs_dir', type=str, default='data_dfs')
    parser.add_argument('--batch_size', type=str)
    parser.add_argument('--torch_device', type=str, default='cuda')
    parser.add_argument('--num_workers', type=str, default='8')
    parser.add_argument('--dropout_rate', type=str, default='0.5')
    parser.add_argument('--gradient_accumulation_steps', type=str, default='1')
    parser.add_argument('--log_frequency', type=int, default=None, help='override the preset cadence (config.py)')
    parser.add_argument('--max_steps', type=int, default=None)
    parser.add_argument('--seed', type=int, default=0, help='validation-order shuffle (the same on every rank)')
    parser.add_argument('--gpus', type=int, default=None,
                        help='data-parallel ranks on this node; > 1 without a launcher environment starts the ranks itself')
    parser.add_argument('--timeout', type=float, default=None,
                        help='wall-clock limit in seconds of a self-launched --gpus N job (default: none; training runs for hours)')
    parser.add_argument('--feats_manifest_dir', type=str, default=None,
                        help='directory with the {split}_feats.jsonl manifests compute_features.py wrote: train from the stored '
                             'feature matrices instead of re-featurising audio (default: <data_root>/<lhotse_dir>/cutsets if it '
                             'holds such manifests, as the reference reads its stored features from --lhotse_dir)')
    args = parser.parse_args(argv)
    warn_ignored_flags(args, parser)

    config = config_mod.MODEL_MAP[args.config]
    batch_size = int(args.batch_size or config['batch_size'])
    log_frequency = args.log_frequency or config['log_frequency']
    grad_accum = int(args.gradient_accumulation_steps)
    if grad_accum < 1:
        raise SystemExit("--gradient_accumulation_steps must be >= 1")
    if args.gpus is not None and args.gpus > 1 and not parallel.under_launcher():
        # parent launcher: nothing here has touched the GPU yet; start the ranks as fresh children and wait for them
        raise SystemExit(parallel.spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:] if argv is None else list(argv),
                                              timeout=args.timeout))
    rank, world, local = parallel.init_from_env()
    if args.gpus is not None and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher environment says WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("train.py needs an MI355X (the HIP path has no CPU fallback)")
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    os.makedirs(args.checkpoint_dir, exist_ok=True)

    print("Initializing model...")
    model = config['model'](dropout_rate=float(args.dropout_rate), linear_layer_size=config['linear_layer_size'],
                            filter_sizes=config['filter_sizes'])
    model.set_device(device)
    torch_utils.count_parameters(model)
    model.apply(torch_utils.init_weights)
    last = os.path.join(args.checkpoint_dir, 'last.pth.tar')
    if os.path.exists(last):
        torch_utils.load_checkpoint(last, model, map_location=device)
    parallel.broadcast_parameters(model)
    reducer = parallel.GradReducer()

    print("Preparing training set...")
    data_dir = os.path.join(args.data_root, args.data_dfs_dir)
    manifests = find_feats_manifests(args)
    if rank == 0:
        print("Stored features: " + (", ".join(f"{k}: {v}" for k, v in manifests.items()) if manifests else
                                     "none found -- channels are featurised from audio on the GPU"))
    dev_loader = load_data.create_training_dataloader(data_dir, 'dev', shuffle=True, seed=args.seed, batch_size=batch_size,
                                                      audio_root=args.data_root, feats_manifest=manifests.get('dev'))
    train_loader = load_data.create_training_dataloader(data_dir, 'train', batch_size=batch_size, audio_root=args.data_root,
                                                        rank=rank, world=world, store=dev_loader.dataset.store,
                                                        feats_manifest=manifests.get('train'))
    if rank == 0:
        with open(os.path.join(args.checkpoint_dir, 'train_params.csv'), 'w', newline='') as f:
            w = csv.writer(f)
            w.writerow(['train_samples', 'val_samples', 'val_samples_per_log', 'log_freq', 'batchsize'])
            w.writerow([train_loader.sampler.num_cuts, dev_loader.sampler.num_cuts, '', log_frequency, batch_size])
    rows = []
    start = time.time()
    for epoch in range(args.num_epochs):
        t0 = time.time()
        loss = run_epoch(model, train_loader, dev_loader, args.checkpoint_dir, log_frequency, batch_size, rows, reducer,
                         rank=rank, max_steps=args.max_steps, grad_accum=grad_accum)
        torch.cuda.synchronize()
        mins, secs = torch_utils.epoch_time(t0, time.time())
        if rank == 0:
            print(f'Epoch: {epoch + 1:02} | Time: {mins}m {secs}s | summed train loss {loss:.4f}')
    if rank == 0:
        print(f"Ran {args.num_epochs} epochs in {time.time() - start:.2f}s")
        update_metrics_on_disk(os.path.join(args.checkpoint_dir, 'metrics.csv'), rows)


if __name__ == '__main__':
    main()
