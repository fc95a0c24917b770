"""Segment index: the table that says which frames of which channel form a training example, and its label.

Reference: compute_features.py:114-195 (`compute_features_for_cuts`: one cut per data-frame row =
`track.truncate(offset=sub_start, duration=sub_duration).pad(duration=1.0)` + supervision `is_laugh = label`, then
`CutSet.shuffle()`), compute_features.py:197-261 (`compute_features_for_single_audio_track`: consecutive 1 s windows,
last partial window dropped), analysis/utils.py:8-15 (`to_frames`: round(t * 1000 / frame_duration)).
Row schema (create_data_df.py:171-172, data/icsi/data_dfs/samples/*.csv):
    start, duration, sub_start, sub_duration, audio_path, meeting_id, chan_id, label

Everything here is integer bookkeeping on the host (bit-exact by construction, tests/test_host_logic.py, tests/test_properties.py); the frames
themselves are gathered on the GPU (csrc/gather.hip).
"""
import csv
from dataclasses import dataclass

import numpy as np

FRAME_SHIFT = 0.01
MIN_SEG_DURATION = 1.0


def seconds_to_frame(t, frame_shift=FRAME_SHIFT):
    """Frame index of a time stamp: round-half-even of t / frame_shift on the 2-decimal values of the data frames
    (create_data_df.py:182 rounds every time to 2 decimals, so t / 0.01 is an integer up to float noise)."""
    return int(round(round(float(t), 2) / frame_shift))


@dataclass
class SegmentTable:
    channel: np.ndarray      # int32 (N,)  index into `channels`
    first_frame: np.ndarray  # int64 (N,)
    n_frames: np.ndarray     # int32 (N,)  <= frames_per_segment; the rest is padding
    label: np.ndarray        # int32 (N,)
    channels: list           # channel keys, e.g. "Bmr021/chan3.sph"
    frames_per_segment: int = 100

    def __len__(self):
        return len(self.label)

    def shuffled(self, seed=None):
        rng = np.random.default_rng(seed)
        perm = rng.permutation(len(self))
        return SegmentTable(self.channel[perm], self.first_frame[perm], self.n_frames[perm], self.label[perm],
                            self.channels, self.frames_per_segment)

    def shard(self, rank, world):
        per = (len(self) + world - 1) // world
        sl = slice(min(len(self), rank * per), min(len(self), (rank + 1) * per))
        return SegmentTable(self.channel[sl], self.first_frame[sl], self.n_frames[sl], self.label[sl], self.channels,
                            self.frames_per_segment)


def table_from_rows(rows, min_seg_duration=MIN_SEG_DURATION, frame_shift=FRAME_SHIFT):
    """rows: iterable of dicts with the data-frame columns -> SegmentTable (one segment per row, row order kept)."""
    fps = int(round(min_seg_duration / frame_shift))
    chans, chan_idx = [], {}
    channel, first, count, label = [], [], [], []
    for r in rows:
        key = r["audio_path"]
        if key not in chan_idx:
            chan_idx[key] = len(chans)
            chans.append(key)
        sub_start, sub_dur = float(r["sub_start"]), float(r["sub_duration"])
        if sub_start < 0 or sub_dur < 0:
            raise ValueError(f"negative time in segment row {r}")  # create_data_df.py:185-186 asserts the same
        lab = int(r["label"])
        if lab not in (0, 1):
            raise ValueError(f"label must be 0 or 1, got {lab}")   # create_data_df.py:189-190
        channel.append(chan_idx[key])
        first.append(seconds_to_frame(sub_start, frame_shift))
        count.append(min(fps, seconds_to_frame(sub_dur, frame_shift)))
        label.append(lab)
    return SegmentTable(np.asarray(channel, np.int32), np.asarray(first, np.int64), np.asarray(count, np.int32),
                        np.asarray(label, np.int32), chans, fps)


def table_from_csv(path, **kw):
    with open(path, newline="") as f:
        return table_from_rows(list(csv.DictReader(f)), **kw)


def whole_track_table(n_frames_total, channel_key, laugh_intervals_ms=(), min_seg_duration=MIN_SEG_DURATION,
                      frame_shift=FRAME_SHIFT):
    """Consecutive windows over one channel (compute_features.py:228-243): window w = [w, w+1) s; the last partial
    window is dropped; label 1 iff the open-closed interval (start_ms, end_ms] overlaps a laugh interval of the
    channel's participant.  laugh_intervals_ms: iterable of (lo_ms, hi_ms] integer pairs (the `portion` index of
    analysis/preprocess.py, in 1 ms frames)."""
    fps = int(round(min_seg_duration / frame_shift))
    n_win = n_frames_total // fps
    first = np.arange(n_win, dtype=np.int64) * fps
    label = np.zeros(n_win, np.int32)
    for w in range(n_win):
        lo, hi = int(round(w * min_seg_duration * 1000)), int(round((w + 1) * min_seg_duration * 1000))
        for a, b in laugh_intervals_ms:  # (lo, hi] overlaps (a, b]  <=>  max(lo, a) < min(hi, b)
            if max(lo, a) < min(hi, b):
                label[w] = 1
                break
    return SegmentTable(np.zeros(n_win, np.int32), first, np.full(n_win, fps, np.int32), label, [channel_key], fps)
