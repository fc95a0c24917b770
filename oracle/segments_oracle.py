"""CPU restatement of the reference's segment-index logic (pure-Python loops; small tables only).

TEST INFRASTRUCTURE ONLY.  Follows compute_features.py:156-195 (one cut per data-frame row: truncate(offset=sub_start,
duration=sub_duration).pad(duration=1.0), supervision is_laugh = row.label) and compute_features.py:228-243 (whole
track in 1 s windows, last partial window dropped, label by overlap of (start_ms, end_ms] with the laugh index;
analysis/utils.py:8-15 to_frames = round(t * 1000)).
The frame arithmetic of `truncate`/`pad` lives in lhotse@f1b66b8a (absent): first frame = round(sub_start / 0.01),
count = round(sub_duration / 0.01) capped at 100 is the published behaviour restated -> **parity unpinned** for the
rounding rule itself; pinned here are the row schema and values of the reference's own sample tables
(tests/golden/data_dfs/*.csv, copied data files) and the window/label logic.
"""


def rows_to_segments(rows, fps=100):
    chans, out = [], []
    for r in rows:
        if r["audio_path"] not in chans:
            chans.append(r["audio_path"])
        first = int(round(round(float(r["sub_start"]), 2) * fps))
        count = min(fps, int(round(round(float(r["sub_duration"]), 2) * fps)))
        out.append((chans.index(r["audio_path"]), first, count, int(r["label"])))
    return chans, out


def whole_track_windows(n_frames_total, laugh_intervals_ms, fps=100):
    out = []
    w = 0
    while (w + 1) * fps <= n_frames_total:
        lo, hi = w * 1000, (w + 1) * 1000
        label = 0
        for a, b in laugh_intervals_ms:
            # portion: openclosed(lo, hi).overlaps(openclosed(a, b))
            if max(lo, a) < min(hi, b):
                label = 1
        out.append((w * fps, fps, label))
        w += 1
    return out
