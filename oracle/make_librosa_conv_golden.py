"""Golden vectors for the librosa-convention half of oracle/fbank_oracle.py (TEST INFRASTRUCTURE).

librosa itself is absent from the image and the reference holds no librosa FEATURE output at all
(/root/reference/segment_laughter.py:134 calls `librosa.load` only), so this half of the oracle
cannot be pinned by the reference.  What the image does hold are two independent third-party
implementations of the same published conventions:

  * `transformers.audio_utils` (`mel_filter_bank(norm="slaney", mel_scale="slaney")`, `spectrogram(center=True,
    pad_mode="reflect", power=2.0)`, `power_to_db`) -- the numpy feature code behind WhisperFeatureExtractor,
    written to reproduce `librosa.filters.mel`, `librosa.stft` / `melspectrogram` and `librosa.power_to_db`;
  * `scipy.fft.dct(type=2, norm="ortho")` -- the very call `librosa.feature.mfcc` makes.

This script runs them on seeded clips and stores inputs and outputs in tests/golden/librosa_conv.npz.
Nothing here reads /root/reference.  The pin is "an independent implementation of the librosa
convention", not librosa: README / DESIGN say so next to the 1e-4 claim.

    python oracle/make_librosa_conv_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

SR, N_FFT, WIN, HOP = 16000, 512, 400, 160


def make_inputs():
    """Five seeded clips: noise, a chirp, speech-like AM noise, near-silence, and a ragged length."""
    rng = np.random.default_rng(20261005)
    t = np.arange(SR) / SR
    clips = [
        (0.1 * rng.standard_normal(SR)).astype(np.float32),
        (0.5 * np.sin(2 * np.pi * (100.0 + 3500.0 * t) * t)).astype(np.float32),
        (0.3 * rng.standard_normal(SR) * (0.5 + 0.5 * np.sin(2 * np.pi * 4.0 * t)) ** 2).astype(np.float32),
        (1e-6 * rng.standard_normal(SR)).astype(np.float32),
    ]
    ragged = (0.2 * rng.standard_normal(12345)).astype(np.float32)
    return np.stack(clips), ragged


def third_party(x, n_mels, pad_mode):
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function
    fb = mel_filter_bank(num_frequency_bins=N_FFT // 2 + 1, num_mel_filters=n_mels, min_frequency=0.0,
                         max_frequency=SR / 2.0, sampling_rate=SR, norm="slaney", mel_scale="slaney")
    win = window_function(WIN, "hann", periodic=True)
    mel = spectrogram(np.asarray(x, np.float64), win, frame_length=WIN, hop_length=HOP, fft_length=N_FFT, power=2.0,
                      center=True, pad_mode=pad_mode, mel_filters=fb, mel_floor=0.0)
    return fb, mel.T.copy()  # (T, n_mels)


def main():
    import scipy
    import transformers
    from scipy.fft import dct
    from transformers.audio_utils import power_to_db

    clips, ragged = make_inputs()
    out = {"clips": clips, "ragged": ragged,
           "versions": np.array([f"transformers {transformers.__version__}", f"scipy {scipy.__version__}",
                                 f"numpy {np.__version__}"])}
    for n_mels in (44, 40, 128):
        fb, _ = third_party(clips[0], n_mels, "reflect")
        out[f"bank_{n_mels}"] = fb
    for pad in ("reflect", "constant"):
        mel = np.stack([third_party(c, 44, pad)[1] for c in clips])
        out[f"mel44_{pad}"] = mel
    out["mel44_ragged"] = third_party(ragged, 44, "reflect")[1]
    mel = out["mel44_reflect"]
    out["db44"] = np.stack([power_to_db(m, reference=1.0, min_value=1e-10, db_range=80.0) for m in mel])
    db_nocut = 10.0 * np.log10(np.maximum(mel, 1e-10))
    out["mfcc20_of_db"] = dct(db_nocut, type=2, norm="ortho", axis=-1)[..., :20]
    out["mfcc13_of_db_top80"] = dct(out["db44"], type=2, norm="ortho", axis=-1)[..., :13]
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "librosa_conv.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
