"""Drop-in for the reference's load_data.py: same two factory functions, data resident on the MI355X.

Reference: load_data.py:12-34 `create_training_dataloader(cutset_dir, split, shuffle=False)` (Lhotse CutSet +
SingleCutSampler(max_cuts=32) + DataLoader(batch_size=None); the loop reads `.sampler.num_cuts`, train.py:305-306),
load_data.py:37-54 `create_inference_dataloader(audio_path)` (whole file -> features -> InferenceDataset ->
DataLoader(batch_size=32)).

Lhotse manifests are replaced by the data-frame CSVs the reference itself produces (create_data_df.py:171-172 schema,
`{split}_df.csv`) plus the audio files they name: channels are featurised once on the GPU and kept in HBM.
Audio: 16 kHz mono .wav (int16/float) or .npy float arrays; NIST .sph needs external conversion (sph2pipe), as in
the reference's own tooling (analysis/output_processing/laughs_to_wav.py).
"""
import os
import sys

import numpy as np
import torch

import config as cfg
from datasets import FeatureStore, InferenceDataset, LadDataset
import segments

sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), "utils"))
from utils import get_feat_extractor  # noqa: E402


def load_audio(path, sampling_rate=16000):
    """Mono float32 in [-1, 1] at `sampling_rate`."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        return np.load(path).astype(np.float32).reshape(-1)
    if ext == ".wav":
        from scipy.io import wavfile
        sr, x = wavfile.read(path)
        if sr != sampling_rate:
            raise ValueError(f"{path}: expected {sampling_rate} Hz audio, got {sr}")
        if x.ndim > 1:
            x = x[:, 0]
        if x.dtype == np.int16:
            return (x.astype(np.float32) / 32768.0)
        if x.dtype == np.int32:
            return (x.astype(np.float32) / 2147483648.0)
        return x.astype(np.float32)
    raise ValueError(f"unsupported audio format {ext!r} ({path}): convert NIST sphere files with sph2pipe first")


class SegmentSampler:
    """Index batches of at most `max_cuts` segments (SingleCutSampler(max_cuts=32), load_data.py:32)."""

    def __init__(self, n, max_cuts=32, rank=0, world=1):
        self.n, self.max_cuts = n, max_cuts
        self.num_cuts = n
        self.rank, self.world = rank, world

    def __iter__(self):
        per = (self.n + self.world - 1) // self.world
        lo, hi = min(self.n, self.rank * per), min(self.n, (self.rank + 1) * per)
        for s in range(lo, hi, self.max_cuts):
            yield np.arange(s, min(hi, s + self.max_cuts))

    def __len__(self):
        per = (self.n + self.world - 1) // self.world
        return (per + self.max_cuts - 1) // self.max_cuts


class SegmentLoader:
    """Iterable of batch dicts with the attributes the reference loop reads (`.sampler.num_cuts`, `.dataset`)."""

    def __init__(self, dataset, sampler):
        self.dataset, self.sampler = dataset, sampler

    def __iter__(self):
        for idx in self.sampler:
            yield self.dataset[idx]

    def __len__(self):
        return len(self.sampler)


def create_training_dataloader(cutset_dir, split, shuffle=False, batch_size=32, audio_root=None, seed=None, rank=0,
                               world=1, store=None):
    '''
    Create a dataloader for the provided split
        - split needs to be one of 'train', 'dev' and 'test'
        - cutset_dir holds `{split}_df.csv` (segment rows) ; audio paths are resolved against `audio_root`
        - shuffle shuffles the segment table before batching (CutSet.shuffle(), load_data.py:27-28)
    '''
    if split not in ['train', 'dev', 'test']:
        raise ValueError(
            f"Unexpected value for split. Needs to be one of 'train, dev, test'. Found {split}")
    table = segments.table_from_csv(os.path.join(cutset_dir, f'{split}_df.csv'))
    if shuffle:
        table = table.shuffled(seed)
    if store is None:
        extractor = get_feat_extractor(num_samples=cfg.FEAT['num_samples'], num_filters=cfg.FEAT['num_filters'])
        store = FeatureStore(extractor)
    root = audio_root if audio_root is not None else cutset_dir
    for key in table.channels:
        if key not in store.keys:
            path = key if os.path.isabs(key) else os.path.join(root, key)
            if not os.path.exists(path):
                for alt in (".wav", ".npy"):
                    if os.path.exists(os.path.splitext(path)[0] + alt):
                        path = os.path.splitext(path)[0] + alt
                        break
            store.add_audio(key, load_audio(path))
    dataset = LadDataset(store, table)
    return SegmentLoader(dataset, SegmentSampler(len(table), max_cuts=batch_size, rank=rank, world=world))


class InferenceLoader:
    """Batches of <= batch_size stride-one-frame windows, (n, 100, F) GPU float32 (DataLoader(batch_size=32))."""

    def __init__(self, dataset, batch_size=32):
        self.dataset, self.batch_size = dataset, batch_size

    def __iter__(self):
        for s in range(0, len(self.dataset), self.batch_size):
            yield self.dataset.batch(s, self.batch_size)

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size


def create_inference_dataloader(audio_path, batch_size=32):
    '''
    Create inference dataloader for the audio file `audio_path`: the whole file is featurised in one GPU launch and the
    windows are read out of that (T, F) matrix.  `loader.dataset.feats` is the matrix itself, for
    `model.engine.predict_windows`, which skips the window materialisation altogether.
    '''
    extractor = get_feat_extractor(num_samples=cfg.FEAT['num_samples'], num_filters=cfg.FEAT['num_filters'])
    pcm = torch.from_numpy(load_audio(audio_path)).to(extractor.config.device)
    feats_all = extractor.extract_long(pcm.contiguous())
    dataset = InferenceDataset(feats_all)
    return InferenceLoader(dataset, batch_size=batch_size)
