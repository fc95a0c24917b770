#!/usr/bin/env python3
"""Offline whole-channel featurisation on the MI355X: counterpart of the reference's compute_features.py stage 1.

Reference: compute_features.py:66-111 `compute_features_per_split` (per split: every recording -> Lhotse
`compute_and_store_features(extractor, storage_path, num_jobs)` = a CPU process pool + lossy lilcom storage + a
`{split}_feats.jsonl` manifest).  Here one recording is one kernel launch (an hour of audio takes a few ms), results
are stored as plain float32 `.npy` matrices (T, 44) with a JSONL manifest, and with torchrun the recordings of a split
are sharded over ranks (they are independent: no collective).  The segment index built by stage 2
(compute_features.py:114-261) is `segments.py`; training reads audio directly (`load_data.create_training_dataloader`),
so this script is only needed when features are to be kept on disk.
"""
import argparse
import json
import os
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(_PKG, "utils"), _PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import config as cfg  # noqa: E402
import load_data  # noqa: E402
import parallel  # noqa: E402
from utils import get_feat_extractor  # noqa: E402

SPLITS = ['train', 'dev', 'test']


def compute_features_per_split(split_audio, output_dir, rank=0, world=1, use_kaldi=False):
    """split_audio: {split: [audio paths]} -> writes <output_dir>/feats/<split>/<id>.npy and
    <output_dir>/cutsets/{split}_feats.jsonl (one record per recording: id, path, num_frames, num_features, frame_shift, extractor configuration)."""
    extractor = get_feat_extractor(num_samples=cfg.FEAT['num_samples'], num_filters=cfg.FEAT['num_filters'], use_kaldi=use_kaldi)
    os.makedirs(os.path.join(output_dir, 'cutsets'), exist_ok=True)
    written = {}
    for split, paths in split_audio.items():
        feats_dir = os.path.join(output_dir, 'feats', split)
        os.makedirs(feats_dir, exist_ok=True)
        records = []
        for i in parallel.shard_indices(len(paths), rank, world):
            path = paths[i]
            rec_id = os.path.splitext(os.path.relpath(path, os.path.commonpath(paths) if len(paths) > 1 else os.path.dirname(path)))[0]
            rec_id = rec_id.replace(os.sep, '_')
            pcm = torch.from_numpy(load_data.load_audio(path)).to(extractor.config.device)
            feats = extractor.extract_long(pcm.contiguous()).cpu().numpy()
            out = os.path.join(feats_dir, rec_id + '.npy')
            np.save(out, feats)
            records.append({'id': rec_id, 'audio_path': path, 'features_path': out, 'num_frames': int(feats.shape[0]),
                            'num_features': int(feats.shape[1]), 'frame_shift': extractor.frame_shift,
                            # what produced the matrix: a loader refuses stored features of another extractor (load_data._stored_features)
                            'extractor': load_data.extractor_signature(extractor)})
        manifest = os.path.join(output_dir, 'cutsets', f'{split}_feats.jsonl' if world == 1 else f'{split}_feats.rank{rank}.jsonl')
        with open(manifest, 'w') as f:
            for r in records:
                f.write(json.dumps(r) + '\n')
        written[split] = records
    return written


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--audio_root', required=True, help='directory with one sub-directory per split holding .wav / .npy audio')
    ap.add_argument('--output_dir', required=True)
    args = ap.parse_args(argv)
    rank, world, local = parallel.init_from_env()
    torch.cuda.set_device(local)
    split_audio = {}
    for split in SPLITS:
        d = os.path.join(args.audio_root, split)
        if os.path.isdir(d):
            split_audio[split] = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith(('.wav', '.npy')))
    out = compute_features_per_split(split_audio, args.output_dir, rank, world)
    if rank == 0:
        for split, recs in out.items():
            print(f'{split}: {len(recs)} recordings, {sum(r["num_frames"] for r in recs)} frames')


if __name__ == '__main__':
    main()
