"""Drop-in for the reference's datasets.py on HBM-resident features.

Reference: datasets.py:23-68 `LadDataset` (item = {'inputs': (B,T,F) f32, 'input_lens': (B,), 'is_laugh': (B,) int32,
'cut': cuts}; built on Lhotse's PrecomputedFeatures: per-cut lilcom read + decompress + pad + stack on the host),
datasets.py:72-93 `InferenceDataset` (item i = feats[i:i+100], zero right-pad at the end of the file).

Here whole-channel feature matrices stay resident in HBM (`FeatureStore`: one hour of audio is 63 MB, the whole ICSI
corpus ~28 GB of the 288 GB) and a batch is one gather launch (csrc/gather.hip) driven by the integer segment table
(segments.py).  There is no lilcom stage: parity is defined against the raw extractor output.
"""
import ctypes

import numpy as np
import torch

import _hip
import config as cfg

# value Lhotse pads short cuts with in the log-mel domain (log(1e-10)); [UPSTREAM-UNVERIFIED], SURVEY.md section 5
LOG_EPSILON = -23.025850929940457


class FeatureStore:
    """Whole-channel (T, F) float32 feature matrices on the GPU + the device-side tables the gather kernel reads."""

    def __init__(self, extractor=None, device="cuda"):
        self.extractor = extractor
        self.device = torch.device(device)
        self.keys, self.mats = [], []
        self._tables = None

    def add_features(self, key, feats):
        f = torch.as_tensor(feats, dtype=torch.float32).to(self.device).contiguous()
        if f.dim() != 2 or f.shape[1] % 4 != 0:
            raise ValueError("features must be (T, F) with F a multiple of 4")
        self.keys.append(key)
        self.mats.append(f)
        self._tables = None
        return len(self.keys) - 1

    def add_audio(self, key, pcm):
        """Featurise one whole channel on the GPU (compute_features.py:66-111, one launch per channel)."""
        if self.extractor is None:
            raise ValueError("FeatureStore needs an extractor to add audio")
        x = torch.as_tensor(np.asarray(pcm) if not isinstance(pcm, torch.Tensor) else pcm).to(self.device, torch.float32)
        return self.add_features(key, self.extractor.extract_long(x.contiguous().view(-1)))

    def index_of(self, key):
        return self.keys.index(key)

    def tables(self):
        if self._tables is None:
            ptrs = torch.tensor([m.data_ptr() for m in self.mats], dtype=torch.int64, device=self.device)
            frames = torch.tensor([m.shape[0] for m in self.mats], dtype=torch.int64, device=self.device)
            self._tables = (ptrs, frames)
        return self._tables

    @property
    def num_filters(self):
        return self.mats[0].shape[1]


def gather_segments(store, chan, first, count, n_frames, pad, out=None):
    """(B, n_frames, F) batch: segment b = frames [first[b], first[b]+count[b]) of channel chan[b], padded with `pad`."""
    ptrs, frames = store.tables()
    n = int(chan.shape[0])
    F = store.num_filters
    if out is None:
        out = torch.empty((n, n_frames, F), device=store.device, dtype=torch.float32)
    _hip.check(_hip.lib().lad_gather_segments(_hip.ptr(ptrs), _hip.ptr(frames), _hip.ptr(chan), _hip.ptr(first), _hip.ptr(count), n,
                                              n_frames, F, float(pad), _hip.ptr(out), _hip.stream_handle(store.device)),
               "lad_gather_segments")
    return out


class LadDataset(torch.utils.data.Dataset):
    """Laugh-activity-detection dataset: indexing with a batch of segment ids returns the reference's batch dict."""

    def __init__(self, store, table, pad_value=LOG_EPSILON):
        super().__init__()
        self.store, self.table, self.pad_value = store, table, pad_value
        remap = np.asarray([store.index_of(k) for k in table.channels], np.int32)
        dev = store.device
        self._chan = torch.from_numpy(remap[table.channel]).to(dev)
        self._first = torch.from_numpy(table.first_frame).to(dev)
        self._count = torch.from_numpy(table.n_frames).to(dev)
        self._label = torch.from_numpy(table.label).to(dev)

    def __len__(self):
        return len(self.table)

    def __getitem__(self, cuts):
        idx = torch.as_tensor(cuts, dtype=torch.int64, device=self.store.device).view(-1)
        chan, first, count = self._chan[idx].contiguous(), self._first[idx].contiguous(), self._count[idx].contiguous()
        inputs = gather_segments(self.store, chan, first, count, self.table.frames_per_segment, self.pad_value)
        return {"inputs": inputs, "input_lens": count, "is_laugh": self._label[idx].contiguous(), "cut": cuts}


class InferenceDataset(torch.utils.data.Dataset):
    """Stride-one-frame windows over the features of a whole file (datasets.py:72-93)."""

    def __init__(self, feats, n_frames=cfg.FEAT['num_samples']) -> None:
        super().__init__()
        self.feats = feats
        self.n_frames = n_frames

    def __len__(self):
        return len(self.feats)

    def __getitem__(self, index):
        ret = self.feats[index:index + self.n_frames]
        if ret.shape[0] != self.n_frames:
            pad_amount = self.n_frames - ret.shape[0]
            if isinstance(ret, torch.Tensor):
                ret = torch.nn.functional.pad(ret, (0, 0, 0, pad_amount))
            else:
                ret = np.pad(ret, ((0, pad_amount), (0, 0)))
        return ret

    def batch(self, start, size):
        """Windows [start, start+size) as one (n, n_frames, F) GPU tensor through the gather kernel."""
        if not (isinstance(self.feats, torch.Tensor) and self.feats.is_cuda):
            raise _hip.LadHipError("InferenceDataset.batch needs GPU-resident features")
        T = self.feats.shape[0]
        n = max(0, min(size, T - start))
        store = FeatureStore(device=self.feats.device)
        store.add_features("file", self.feats)
        dev = self.feats.device
        first = torch.arange(start, start + n, dtype=torch.int64, device=dev)
        count = torch.clamp(T - first, max=self.n_frames).to(torch.int32)
        return gather_segments(store, torch.zeros(n, dtype=torch.int32, device=dev), first, count, self.n_frames, 0.0)
