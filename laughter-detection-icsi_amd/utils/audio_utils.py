"""Drop-in for the reference's utils/audio_utils.py (one function, audio_utils.py:7-9).

The reference asks `audioread` for the duration; here it is read from the container itself: the RIFF header of a
.wav (no sample data is touched) or the shape of a .npy array (memory-mapped), at the 16 kHz the path runs at.
"""
import os
import wave

import numpy as np


def get_audio_length(path, sampling_rate=16000):
    """Duration of the file in seconds (float) = samples / rate, as audioread reports it."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        return np.load(path, mmap_mode="r").size / float(sampling_rate)  # load_data.load_audio flattens it
    if ext == ".wav":
        try:
            with wave.open(path, "rb") as f:
                return f.getnframes() / float(f.getframerate())
        except wave.Error:  # float / extensible wav: fall back to scipy's header parser
            from scipy.io import wavfile
            sr, x = wavfile.read(path, mmap=True)
            return x.shape[0] / float(sr)
    raise ValueError(f"unsupported audio format {ext!r} ({path}): convert NIST sphere files with sph2pipe first")


def get_sampling_rate(path, default=16000):
    """Sampling rate the file declares (.npy arrays carry none: `default`)."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        return default
    if ext == ".wav":
        try:
            with wave.open(path, "rb") as f:
                return f.getframerate()
        except wave.Error:
            from scipy.io import wavfile
            return wavfile.read(path, mmap=True)[0]
    raise ValueError(f"unsupported audio format {ext!r} ({path}): convert NIST sphere files with sph2pipe first")
