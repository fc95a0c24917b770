"""Golden vectors for the Kaldi / Lhotse half of oracle/fbank_oracle.py from a third party (TEST INFRASTRUCTURE).

The reference's extractor is Lhotse's `Fbank` (utils/utils.py:25) = `torchaudio.compliance.kaldi.fbank`; neither lhotse nor
torchaudio is in the image, and the only Lhotse output the reference holds are the two feature PLOTS of Demo.ipynb, which pin
the oracle at 0.02-0.06 ln (tests/golden/demo_fbank_plot.npz).  This script adds a pin at the tolerance the kernel is judged
at: `transformers.audio_utils` carries a numpy port of that very torchaudio function (the fallback of SeamlessM4T's and
AST's feature extractors when torchaudio is missing: `mel_filter_bank(mel_scale="kaldi", triangularize_in_mel_space=True)`,
`window_function("povey")`, `spectrogram(center=False, preemphasis=0.97, remove_dc_offset=True, log_mel="log",
mel_floor=1.192092955078125e-07)`).  It implements snip_edges=True only, so the frames' PLACEMENT (snip_edges=False: frame t
starts at 160 t - 120, both ends mirrored, round(n / 160) frames) is prepared here by mirror-padding the clip -- that rule
stays pinned by the plots alone (they reject a one-frame shift) -- and everything per frame (DC removal, pre-emphasis, Povey
window, 512-point power spectrum, 20 .. 7600 Hz Kaldi mel bank, log floor) comes from the third party.

Inputs: the seeded clips of make_librosa_conv_golden.py and the reference's two demo recordings (tests/golden/demo_clips.npz).
Nothing here reads /root/reference.  Output: tests/golden/kaldi_conv.npz.

    python oracle/make_kaldi_conv_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

WIN, HOP, N_FFT = 400, 160, 512


def mirror_padded(x):
    """snip_edges=False placement: (n + 80) // 160 frames, frame t = samples [160 t - 120, 160 t + 280), ends mirrored
    (x[-1 - i] before the start, x[n - 1 - i] past the end: numpy's "symmetric")."""
    x = np.asarray(x, np.float64)
    t = (len(x) + HOP // 2) // HOP
    left = (WIN - HOP) // 2
    right = (t - 1) * HOP + WIN - len(x) - left
    return np.pad(x, (left, max(right, 0)), mode="symmetric"), t


def third_party_fbank(x, num_filters):
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function
    fb = mel_filter_bank(num_frequency_bins=N_FFT // 2 + 1, num_mel_filters=num_filters, min_frequency=20.0,
                         max_frequency=8000.0 - 400.0, sampling_rate=16000, norm=None, mel_scale="kaldi",
                         triangularize_in_mel_space=True)
    win = window_function(WIN, "povey", periodic=False)
    ext, t = mirror_padded(x)
    s = spectrogram(ext, win, frame_length=WIN, hop_length=HOP, fft_length=N_FFT, power=2.0, center=False, preemphasis=0.97,
                    mel_filters=fb, log_mel="log", mel_floor=1.192092955078125e-07, remove_dc_offset=True, dtype=np.float64)
    assert s.shape == (num_filters, t), (s.shape, t)
    return fb, s.T.copy()


def main():
    import transformers
    from oracle.make_librosa_conv_golden import make_inputs
    clips, ragged = make_inputs()
    demo = np.load(os.path.join(ROOT, "tests", "golden", "demo_clips.npz"))
    out = {"versions": np.array([f"transformers {transformers.__version__}", f"numpy {np.__version__}"])}
    out["bank_44"], _ = third_party_fbank(clips[0], 44)
    out["bank_40"], _ = third_party_fbank(clips[0], 40)
    out["fbank44_clips"] = np.stack([third_party_fbank(c, 44)[1] for c in clips])
    out["fbank44_ragged"] = third_party_fbank(ragged, 44)[1]
    for name in ("clip0", "clip1"):
        x = demo[name].astype(np.float32) / 32768.0
        out[f"fbank40_demo_{name}"] = third_party_fbank(x, 40)[1]
        out[f"fbank44_demo_{name}"] = third_party_fbank(x, 44)[1]
    path = os.path.join(ROOT, "tests", "golden", "kaldi_conv.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
