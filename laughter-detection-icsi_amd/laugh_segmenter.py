"""Drop-in for the live part of the reference's laugh_segmenter.py: probability track -> laughter instances.

Reference: laugh_segmenter.py:57-71 (fix_over_underflow), :19-24 (collapse / frame_span_to_time_span), :35-42
(cut_laughter_segments), :74-111 (get_laughter_instances), :141-149 (format_outputs).  Same function names, arguments and result
({(threshold, min_length): [(start_s, end_s), ...]}); the per-frame Python loop (87 passes over 360,000 frames in
the evaluation sweeps, cluster_scripts/gen_eval_exp.py:30-36) is replaced by one vectorised run-length pass per
threshold.  Integer run boundaries are bit-exact with the reference (tests/test_host_logic.py against vectors produced
by the reference itself); this is host-side integer bookkeeping on a (T,) vector, not part of the GPU arithmetic.
The Gillick-era MFCC code below laugh_segmenter.py:115 is dead in the reference and is not reproduced.
"""
import numpy as np


def frame_span_to_time_span(frame_span, fps=100.):
    return (frame_span[0] / fps, frame_span[1] / fps)


def collapse_to_start_and_end_frame(instance_list):
    return (instance_list[0], instance_list[-1])


def seconds_to_samples(s, sr):
    return s * sr


def cut_laughter_segments(instance_list, y, sr):
    new_audio = []
    for start, end in instance_list:
        clip = y[int(seconds_to_samples(start, sr)):int(seconds_to_samples(end, sr))]
        new_audio = np.concatenate([new_audio, clip])
    return new_audio


def fix_over_underflow(prob):
    """p > 1 -> 1; p <= 0 -> 1e-7 (so that threshold 0 still accepts the frame); else p."""
    if prob > 1:
        return 1
    if prob <= 0:
        return 0.0000001
    return prob


def fix_probs(probs):
    """Vector form of fix_over_underflow in float64 (the reference maps Python floats)."""
    p = np.asarray(probs, dtype=np.float64).copy()
    p[p > 1] = 1.0
    p[p <= 0] = 0.0000001
    return p


def run_spans(mask):
    """Maximal runs of True in a boolean vector -> int64 array (n_runs, 2) of (first_frame, last_frame)."""
    m = np.asarray(mask, dtype=bool)
    if m.size == 0:
        return np.zeros((0, 2), np.int64)
    d = np.diff(m.astype(np.int8))
    starts = np.flatnonzero(d == 1) + 1
    ends = np.flatnonzero(d == -1)
    if m[0]:
        starts = np.concatenate([[0], starts])
    if m[-1]:
        ends = np.concatenate([ends, [m.size - 1]])
    return np.stack([starts, ends], axis=1).astype(np.int64)


def get_laughter_instances(probs, thresholds=[0.5], min_lengths=[0.2], fps=100.):
    """{(threshold, min_length): [(start_s, end_s), ...]} exactly as laugh_segmenter.py:74-111:
    frame i is laughter iff p[i] > threshold; maximal runs -> (first/fps, last/fps); kept iff end - start > min_length."""
    p = fix_probs(probs)
    instance_dict = {}
    for thr in thresholds:
        spans = run_spans(p > thr)
        inst_all = [(int(a) / fps, int(b) / fps) for a, b in spans]
        for min_l in min_lengths:
            instance_dict[(thr, min_l)] = [inst for inst in inst_all if inst[1] - inst[0] > min_l]
    # the reference iterates thresholds-major, min_lengths-minor: restore that key order
    return {(thr, min_l): instance_dict[(thr, min_l)] for thr in thresholds for min_l in min_lengths}


def format_outputs(instances, wav_paths=None):
    """[{'start', 'end'}] (+ 'filename' when the instances were cut into wav files): laugh_segmenter.py:141-149, what
    segment_laughter.py:148 prints after writing `laugh_<i>.wav`."""
    outs = []
    for i, inst in enumerate(instances):
        d = {'start': inst[0], 'end': inst[1]}
        if wav_paths is not None:
            d = {'filename': wav_paths[i], **d}   # IndexError on a short list, as the reference
        outs.append(d)
    return outs


def get_laughter_frame_spans(probs, threshold):
    """Integer (first_frame, last_frame) runs for one threshold: the bit-exact core of get_laughter_instances."""
    return run_spans(fix_probs(probs) > threshold)
