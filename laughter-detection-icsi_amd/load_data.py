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
import parallel
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
    """Index batches of at most `max_cuts` segments (SingleCutSampler(max_cuts=32), load_data.py:32).

    With world > 1 the segments are dealt round-robin to the ranks and the number of batches is the same on every rank
    (parallel.train_step_count): each training step issues one all-reduce, so ranks must never disagree on the count.
    A ragged last batch is dropped on ALL ranks when any rank would see fewer than `min_batch` segments in it."""

    def __init__(self, n, max_cuts=32, rank=0, world=1, min_batch=1):
        self.n, self.max_cuts = n, max_cuts
        self.num_cuts = n
        self.rank, self.world, self.min_batch = rank, world, min_batch

    def __iter__(self):
        return parallel.train_batches(self.n, self.max_cuts, self.rank, self.world, self.min_batch)

    def __len__(self):
        return parallel.train_step_count(self.n, self.max_cuts, self.world, self.min_batch)


class SegmentLoader:
    """Iterable of batch dicts with the attributes the reference loop reads (`.sampler.num_cuts`, `.dataset`)."""

    def __init__(self, dataset, sampler):
        self.dataset, self.sampler = dataset, sampler

    def __iter__(self):
        for idx in self.sampler:
            yield self.dataset[idx]

    def __len__(self):
        return len(self.sampler)


INDEX_SHUFFLE_SEED = 0


def load_segment_table(cutset_dir, split, shuffle=False, seed=None, world=1, index_seed=INDEX_SHUFFLE_SEED):
    """The epoch's segment order for `split` (host integers only, no GPU): `{split}_df.csv` -> index-time permutation
    (compute_features.py:191-193) -> optional loader-time shuffle (load_data.py:27-28).  Identical on every rank."""
    if split not in ['train', 'dev', 'test']:
        raise ValueError(
            f"Unexpected value for split. Needs to be one of 'train, dev, test'. Found {split}")
    table = segments.table_from_csv(os.path.join(cutset_dir, f'{split}_df.csv'))
    if index_seed is not None:
        table = table.shuffled(index_seed)
    if shuffle:
        if seed is None and world > 1:
            raise ValueError("shuffle=True in a data-parallel job needs an explicit seed (every rank must draw the same order)")
        table = table.shuffled(seed)
    return table


def read_feats_manifest(path):
    """`{split}_feats.jsonl` of compute_features.compute_features_per_split -> {key: record}; a recording is found by the
    absolute path of its audio file, by that path without its extension, or by its id."""
    import json
    by_key = {}
    if os.path.isdir(path):
        files = sorted(os.path.join(path, f) for f in os.listdir(path) if f.endswith('.jsonl'))
    else:
        # a data-parallel compute_features run leaves one manifest per rank next to the name asked for
        stem = path[:-len('.jsonl')] if path.endswith('.jsonl') else path
        d = os.path.dirname(path) or '.'
        files = [path] if os.path.isfile(path) else []
        files += sorted(os.path.join(d, f) for f in os.listdir(d)
                        if f.startswith(os.path.basename(stem) + '.rank') and f.endswith('.jsonl'))
    if not files:
        raise FileNotFoundError(f"no feature manifest at {path}")
    for fpath in files:
        with open(fpath) as f:
            for line in f:
                line = line.strip()
                if not line:
                    continue
                rec = json.loads(line)
                ap = os.path.abspath(rec['audio_path'])
                by_key[ap] = rec
                by_key[os.path.splitext(ap)[0]] = rec
                by_key.setdefault('id:' + rec['id'], rec)
    return by_key


def extractor_signature(extractor):
    """What decides the numbers an extractor writes: its configuration without the device (feats.HipFbankConfig).  Stored with
    every manifest record by compute_features.py and compared here before stored features are trained on."""
    d = dict(extractor.config.to_dict())
    d.pop('device', None)
    return d


def _stored_features(manifest, path, num_filters, extractor=None):
    """The stored (T, F) matrix of the recording whose audio is `path`, or None when the manifest does not list it.
    extractor: the configured feature extractor; stored features that another configuration produced (frame shift, filter
    bank, window, ...) are refused -- they would be trained on silently otherwise."""
    ap = os.path.abspath(path)
    rec = manifest.get(ap) or manifest.get(os.path.splitext(ap)[0])
    if rec is None:
        return None
    if extractor is not None:
        want = extractor_signature(extractor)
        have = rec.get('extractor')
        if have is not None:
            diff = sorted(k for k in set(want) | set(have) if want.get(k) != have.get(k))
            if diff:
                raise ValueError(f"{rec['features_path']}: stored features come from another extractor configuration (differs in "
                                 f"{', '.join(diff)}); re-run compute_features.py or train from the audio")
        elif abs(float(rec.get('frame_shift', want['frame_shift'])) - float(want['frame_shift'])) > 1e-12:
            raise ValueError(f"{rec['features_path']}: stored features have frame shift {rec['frame_shift']}, the configured extractor "
                             f"{want['frame_shift']}")
    feats = np.load(rec['features_path'])
    if feats.ndim != 2 or feats.shape[0] != rec['num_frames'] or feats.shape[1] != num_filters:
        raise ValueError(f"{rec['features_path']}: stored features {feats.shape} do not match the manifest "
                         f"({rec['num_frames']}, {rec['num_features']}) / the configured {num_filters} filters")
    return feats.astype(np.float32, copy=False)


def create_training_dataloader(cutset_dir, split, shuffle=False, batch_size=32, audio_root=None, seed=None, rank=0,
                               world=1, store=None, index_seed=INDEX_SHUFFLE_SEED, feats_manifest=None):
    '''
    Create a dataloader for the provided split
        - split needs to be one of 'train', 'dev' and 'test'
        - cutset_dir holds `{split}_df.csv` (segment rows) ; audio paths are resolved against `audio_root`
        - shuffle shuffles the segment table before batching (CutSet.shuffle(), load_data.py:27-28)
        - index_seed: the data-frame CSVs list every speech row, then every laugh row (create_data_df.py:177-179); the
          reference mixes them ONCE when it builds the segment index (`cuts.shuffle()`, compute_features.py:191-193) and
          its loaders then read that stored order.  The same one-off permutation is applied here, with a fixed seed so
          that every rank of a data-parallel job derives the same order BEFORE the segments are dealt to the ranks.
          None keeps the CSV order (single-class batches on the reference's tables: only for tests).
        - feats_manifest: a `{split}_feats.jsonl` written by compute_features.compute_features_per_split (or a directory of
          them).  Channels it lists are loaded from their stored (T, F) matrices instead of being featurised again -- the
          reference trains from stored features too (compute_features.py:105-111 writes them, load_data.py:24-25 and
          datasets.py:56 read them).  The stored matrices are the extractor's raw float32 output (no lilcom stage), so the
          batches are bit-equal to those of the audio path; channels the manifest does not list fall back to their audio.
    '''
    table = load_segment_table(cutset_dir, split, shuffle=shuffle, seed=seed, world=world, index_seed=index_seed)
    if store is None:
        extractor = get_feat_extractor(num_samples=cfg.FEAT['num_samples'], num_filters=cfg.FEAT['num_filters'])
        store = FeatureStore(extractor)
    root = audio_root if audio_root is not None else cutset_dir
    manifest = read_feats_manifest(feats_manifest) if feats_manifest is not None else None
    for key in table.channels:
        if key not in store.keys:
            path = key if os.path.isabs(key) else os.path.join(root, key)
            if not os.path.exists(path):
                for alt in (".wav", ".npy"):
                    if os.path.exists(os.path.splitext(path)[0] + alt):
                        path = os.path.splitext(path)[0] + alt
                        break
            stored = _stored_features(manifest, path, cfg.FEAT['num_filters'], getattr(store, 'extractor', None)) if manifest is not None else None
            if stored is not None:
                store.add_features(key, stored)
            else:
                store.add_audio(key, load_audio(path))
    dataset = LadDataset(store, table)
    # train-mode BatchNorm needs two segments per batch on every rank: a shorter ragged tail is dropped on all ranks alike
    return SegmentLoader(dataset, SegmentSampler(len(table), max_cuts=batch_size, rank=rank, world=world,
                                                 min_batch=2 if split == 'train' else 1))


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
