"""GPU feature extractor with the surface the reference expects from Lhotse's `Fbank`.

`get_feat_extractor` (utils/utils.py) returns a `HipFbank`; the reference uses such an object through
`cut.compute_features(extractor)` (load_data.py:49) and `compute_and_store_features(extractor=...)`
(compute_features.py:105-109), i.e. `.extract(samples, sampling_rate)`, `.frame_shift`,
`.feature_dim(sr)`, `.name`, `.config`.

All arithmetic per frame runs in one HIP kernel (csrc/fbank.hip).  This module only builds the small
host tables the kernel consumes (window, mel filterbank, DCT matrix) -- the convention (Kaldi/Lhotse vs
librosa) is therefore a host-side table choice and does not change the kernel.
"""
import ctypes
import os
from dataclasses import asdict, dataclass

import numpy as np
import torch

import _hip

N_FFT = 512
LN_FLOOR = float(np.finfo(np.float32).eps)


# ------------------------------------------------------------------------------------------ tables
def povey_window(n):
    i = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * i / (n - 1))) ** 0.85


def hann_window_periodic(n):
    i = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * i / n)


def _mel_kaldi(hz):
    return 1127.0 * np.log1p(np.asarray(hz, np.float64) / 700.0)


def kaldi_mel_bank(n_mels, sr, low_freq=20.0, high_freq=-400.0, n_fft=N_FFT, variant="kaldi"):
    """Triangular filters on the 1127*ln(1+f/700) scale, unnormalised -> (n_fft/2+1, n_mels) float64.

    variant "kaldi":   bin j sits at j*sr/n_fft (Kaldi / torchaudio get_mel_banks); Nyquist row unused.
    variant "lhotse0": bin j sits at linspace(0, sr, n_fft)[j] with strict edges (early lhotse create_mel_scale).
    """
    hi = sr / 2.0 + high_freq if high_freq <= 0 else high_freq
    edges = np.linspace(_mel_kaldi(low_freq), _mel_kaldi(hi), n_mels + 2)
    nb = n_fft // 2
    if variant == "kaldi":
        binmel = _mel_kaldi(np.arange(nb) * (sr / n_fft))
    elif variant == "lhotse0":
        binmel = _mel_kaldi(np.linspace(0.0, sr, n_fft)[:nb])
    else:
        raise ValueError(f"unknown mel bank variant {variant!r}")
    bank = np.zeros((nb + 1, n_mels))
    for m in range(n_mels):
        l, c, r = edges[m], edges[m + 1], edges[m + 2]
        rising = (binmel - l) / (c - l)
        falling = (r - binmel) / (r - c)
        tri = np.minimum(rising, falling)
        if variant == "lhotse0":
            tri = np.where((binmel > l) & (binmel < r), np.where(binmel <= c, rising, falling), 0.0)
        bank[:nb, m] = np.clip(tri, 0.0, None)
    return bank


def slaney_mel_bank(n_mels, sr, fmin=0.0, fmax=None, n_fft=N_FFT):
    """librosa.filters.mel defaults (Slaney scale, area-normalised) -> (n_fft/2+1, n_mels) float64."""
    fmax = sr / 2.0 if fmax is None else fmax
    f_sp, brk = 200.0 / 3.0, 1000.0
    step = np.log(6.4) / 27.0

    def to_mel(f):
        f = np.asarray(f, np.float64)
        return np.where(f >= brk, brk / f_sp + np.log(np.maximum(f, 1e-10) / brk) / step, f / f_sp)

    def to_hz(m):
        m = np.asarray(m, np.float64)
        return np.where(m >= brk / f_sp, brk * np.exp(step * (m - brk / f_sp)), f_sp * m)

    pts = to_hz(np.linspace(to_mel(fmin), to_mel(fmax), n_mels + 2))
    freqs = np.arange(n_fft // 2 + 1) * (sr / n_fft)
    bank = np.zeros((n_fft // 2 + 1, n_mels))
    for m in range(n_mels):
        up = (freqs - pts[m]) / (pts[m + 1] - pts[m])
        down = (pts[m + 2] - freqs) / (pts[m + 2] - pts[m + 1])
        bank[:, m] = np.clip(np.minimum(up, down), 0.0, None) * (2.0 / (pts[m + 2] - pts[m]))
    return bank


def dct2_ortho(n_in, n_out):
    """Orthonormal DCT-II as an (n_in, n_out) matrix (scipy.fft.dct(type=2, norm='ortho') along the mel axis)."""
    n = np.arange(n_in, dtype=np.float64)[:, None]
    k = np.arange(n_out, dtype=np.float64)[None, :]
    mat = np.sqrt(2.0 / n_in) * np.cos(np.pi * (n + 0.5) * k / n_in)
    mat[:, 0] /= np.sqrt(2.0)
    return mat


# ------------------------------------------------------------------------------------------ config
@dataclass
class HipFbankConfig:
    """Field names follow Lhotse's FbankConfig where they overlap (defaults = Lhotse defaults)."""
    sampling_rate: int = 16000
    frame_length: float = 0.025
    frame_shift: float = 0.01
    remove_dc_offset: bool = True
    preemph_coeff: float = 0.97
    window_type: str = "povey"
    dither: float = 0.0
    snip_edges: bool = False
    low_freq: float = 20.0
    high_freq: float = -400.0
    num_filters: int = 80
    norm_filters: bool = False
    # "kaldi" (bin j at j*sr/n_fft, torchaudio-compatible) | "lhotse0" (early create_mel_scale).  Default by evidence: the
    # reference's own plot_features() pictures (Demo.ipynb) fit "kaldi" on both recordings, max residual 2.7 vs 8.4 and
    # 3.0 vs 4.9 colour levels (tests/test_fbank_gpu.py::test_hip_fbank_matches_the_reference_feature_plots)
    mel_variant: str = "kaldi"
    convention: str = "kaldi"       # "kaldi" (Lhotse Fbank) | "librosa" (melspectrogram/power_to_db/MFCC)
    num_ceps: int = 0               # >0: output the first num_ceps DCT-II coefficients (MFCC)
    pad_mode: str = "reflect"       # librosa convention only: "reflect" | "constant"
    device: str = "cuda"

    def to_dict(self):
        return asdict(self)


class HipFbank:
    """Log-mel / MFCC extractor running on the MI355X through liblad_hip.so."""

    name = "hip-fbank"

    def __init__(self, config=None):
        self.config = config or HipFbankConfig()
        c = self.config
        if c.dither != 0.0:
            raise ValueError("dither is not supported (the reference runs with dither=0.0)")
        if c.snip_edges:
            raise ValueError("snip_edges=True is not supported (the reference runs with snip_edges=False)")
        sr = c.sampling_rate
        win_len = int(round(c.frame_length * sr))
        hop = int(round(c.frame_shift * sr))
        window = np.zeros(N_FFT)
        if c.convention == "kaldi":
            if c.window_type != "povey":
                raise ValueError("only the povey window is implemented for the kaldi convention")
            window[:win_len] = povey_window(win_len)
            bank = kaldi_mel_bank(c.num_filters, sr, c.low_freq, c.high_freq, N_FFT, c.mel_variant)
            if c.norm_filters:
                bank = bank / bank.sum(axis=0, keepdims=True)
            cfg = _hip.FbankCfg(N_FFT, win_len, hop, c.num_filters, c.num_ceps, 0, 0,
                                int(c.remove_dc_offset), float(c.preemph_coeff), LN_FLOOR)
        elif c.convention == "librosa":
            lpad = (N_FFT - win_len) // 2
            window[lpad:lpad + win_len] = hann_window_periodic(win_len)
            bank = slaney_mel_bank(c.num_filters, sr, n_fft=N_FFT)
            pad = {"reflect": 1, "constant": 2}[c.pad_mode]
            cfg = _hip.FbankCfg(N_FFT, N_FFT, hop, c.num_filters, c.num_ceps, pad, 1, 0, 0.0, 1e-10)
        else:
            raise ValueError(f"unknown convention {c.convention!r}")
        self._hop = hop
        self._n_out = c.num_ceps if c.num_ceps > 0 else c.num_filters
        win32 = np.ascontiguousarray(window, np.float32)
        bank32 = np.ascontiguousarray(bank, np.float32)
        dct32 = np.ascontiguousarray(dct2_ortho(c.num_filters, c.num_ceps), np.float32) if c.num_ceps > 0 else None
        self._plan = ctypes.c_void_p()
        _hip.check(_hip.lib().lad_fbank_plan_create(
            ctypes.byref(cfg), win32.ctypes.data_as(ctypes.c_void_p), bank32.ctypes.data_as(ctypes.c_void_p),
            dct32.ctypes.data_as(ctypes.c_void_p) if dct32 is not None else None, ctypes.byref(self._plan)),
            "lad_fbank_plan_create")

    def __del__(self):
        plan = getattr(self, "_plan", None)
        if plan is not None and plan.value:
            try:
                _hip.lib().lad_fbank_plan_destroy(plan)
            except Exception:
                pass
            self._plan = None

    # -- kernel choice (tests): the library picks the fast kernel whenever the configuration allows it -------------------
    @property
    def has_fast_kernel(self):
        return bool(_hip.lib().lad_fbank_plan_has_fast_kernel(self._plan))

    def use_general_kernel(self, flag=True):
        """Pin the general kernel (csrc/fbank.hip) instead of the 16-lanes-per-frame one (csrc/fbank16.hip)."""
        _hip.check(_hip.lib().lad_fbank_plan_set_kernel(self._plan, 1 if flag else 0), "lad_fbank_plan_set_kernel")
        return self

    # -- Lhotse FeatureExtractor surface ---------------------------------------------------------
    @property
    def frame_shift(self):
        return self.config.frame_shift

    def feature_dim(self, sampling_rate=None):
        return self._n_out

    def num_frames(self, n_samples):
        return int(_hip.lib().lad_fbank_num_frames(self._plan, int(n_samples)))

    def extract(self, samples, sampling_rate=None):
        """samples: np.ndarray / tensor of shape (N,) or (1,N), float in [-1,1] -> np.ndarray (T, F) float32."""
        if sampling_rate is not None and int(sampling_rate) != self.config.sampling_rate:
            raise ValueError(f"extractor is configured for {self.config.sampling_rate} Hz, got {sampling_rate}")
        x = torch.as_tensor(np.asarray(samples) if not isinstance(samples, torch.Tensor) else samples)
        if x.dim() == 2:
            if x.shape[0] != 1:
                raise ValueError("extract() expects mono audio: shape (N,) or (1,N)")
            x = x[0]
        x = x.to(device=self.config.device, dtype=torch.float32).contiguous()
        return self.extract_long(x).cpu().numpy()

    # -- device-resident API (what the fused training / inference loops use) ----------------------------
    def extract_batch(self, pcm, out=None):
        """pcm: GPU float32 (B, N) -> GPU float32 (B, T, F); asynchronous on the current stream."""
        _hip.require_cuda(pcm, "pcm", torch.float32)
        if pcm.dim() != 2:
            raise ValueError("pcm must be (B, N)")
        b, n = pcm.shape
        t = self.num_frames(n)
        if out is None:
            out = torch.empty((b, t, self._n_out), device=pcm.device, dtype=torch.float32)
        else:
            _hip.require_cuda(out, "out", torch.float32)
            if tuple(out.shape) != (b, t, self._n_out):
                raise ValueError(f"out must be {(b, t, self._n_out)}")
        _hip.check(_hip.lib().lad_fbank_forward(self._plan, _hip.ptr(pcm), b, n, _hip.ptr(out),
                                                _hip.stream_handle(pcm.device)), "lad_fbank_forward")
        return out

    def extract_long(self, pcm, out=None):
        """pcm: GPU float32 (N,) -> GPU float32 (T, F): one whole channel (load_data.py:44-49)."""
        _hip.require_cuda(pcm, "pcm", torch.float32)
        if pcm.dim() != 1:
            raise ValueError("pcm must be (N,)")
        n = pcm.shape[0]
        t = self.num_frames(n)
        if out is None:
            out = torch.empty((t, self._n_out), device=pcm.device, dtype=torch.float32)
        _hip.check(_hip.lib().lad_fbank_forward_long(self._plan, _hip.ptr(pcm), n, _hip.ptr(out),
                                                     _hip.stream_handle(pcm.device)), "lad_fbank_forward_long")
        return out


def power_to_db_top(db, top_db=80.0):
    """librosa.power_to_db's final clip (needs the global max, hence outside the per-frame kernel)."""
    return torch.maximum(db, db.max() - top_db)
