"""Property tests (hypothesis) of the host-side integer logic against the pure-Python oracle restatements."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import segmenter_oracle as so, segments_oracle as sgo

probs_st = st.lists(st.one_of(st.floats(-0.5, 1.5, allow_nan=False), st.sampled_from([0.0, 0.5, 1.0])), min_size=0, max_size=300)


@settings(max_examples=150, deadline=None)
@given(probs=probs_st, thr=st.sampled_from([0.0, 0.25, 0.5, 0.75, 1.0]), min_len=st.sampled_from([0.0, 0.05, 0.2]),
       fps=st.sampled_from([100.0, 99.7, 50.0]))
def test_laughter_instances_equal_the_oracle(probs, thr, min_len, fps):
    import laugh_segmenter as ls
    got = ls.get_laughter_instances(probs, [thr], [min_len], fps)
    ref = so.laughter_instances(probs, [thr], [min_len], fps)
    assert got == ref
    spans = ls.get_laughter_frame_spans(probs, thr)
    assert [tuple(int(v) for v in r) for r in spans] == so.run_indices(probs, thr)
    # runs are disjoint, ordered, separated by at least one non-laughter frame
    for (a0, a1), (b0, b1) in zip(spans[:-1], spans[1:]):
        assert a0 <= a1 < b0 - 1 <= b1


@settings(max_examples=100, deadline=None)
@given(rows=st.lists(st.tuples(st.integers(0, 359000), st.integers(1, 100), st.sampled_from(["a.sph", "b.sph", "c/d.sph"]),
                               st.integers(0, 1)), min_size=0, max_size=40))
def test_segment_table_equals_the_oracle(rows):
    import segments
    recs = [dict(start=f0 / 100, duration=n / 100, sub_start=round(f0 / 100, 2), sub_duration=round(n / 100, 2), audio_path=p,
                 meeting_id="m", chan_id="c", label=lab) for f0, n, p, lab in rows]
    t = segments.table_from_rows(recs)
    chans, ref = sgo.rows_to_segments(recs)
    assert t.channels == chans
    assert list(zip(t.channel.tolist(), t.first_frame.tolist(), t.n_frames.tolist(), t.label.tolist())) == ref
    # times with two decimals map to the frame they name
    assert t.first_frame.tolist() == [f0 for f0, _, _, _ in rows]
    assert t.n_frames.tolist() == [n for _, n, _, _ in rows]


@settings(max_examples=60, deadline=None)
@given(n_frames=st.integers(0, 2000), laughs=st.lists(st.tuples(st.integers(0, 20000), st.integers(1, 3000)), max_size=6))
def test_whole_track_windows_equal_the_oracle(n_frames, laughs):
    import segments
    iv = [(a, a + d) for a, d in laughs]
    t = segments.whole_track_table(n_frames, "x", iv)
    ref = sgo.whole_track_windows(n_frames, iv)
    assert list(zip(t.first_frame.tolist(), t.n_frames.tolist(), t.label.tolist())) == ref


@settings(max_examples=60, deadline=None)
@given(n=st.integers(0, 5000), world=st.integers(1, 9))
def test_shards_partition_the_range(n, world):
    import parallel
    seen = []
    for r in range(world):
        seen.extend(parallel.shard_indices(n, r, world))
    assert seen == list(range(n))
