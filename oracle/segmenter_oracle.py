"""CPU restatement of the reference's probability-track segmentation.

TEST INFRASTRUCTURE ONLY.  Pinned by tests/golden/segmenter.json, generated
from the reference's own `laugh_segmenter.get_laughter_instances`
(oracle/make_goldens.py).

Follows laugh_segmenter.py:57-71 (fix_over_underflow), :23-24
(frame_span_to_time_span) and :74-111 (get_laughter_instances):
  * p > 1 -> 1 ; p <= 0 -> 1e-7 ; else p
  * frame i is laughter iff p[i] > thr   (np.min(probs[i:i+1]) is just p[i])
  * maximal runs -> (first/fps, last/fps); keep iff end - start > min_len
  * result keyed (thr, min_len) in thresholds-major order
Pure-Python loops: meant for small tracks only.
"""


def fix_prob(p):
    if p > 1:
        return 1
    if p <= 0:
        return 0.0000001
    return p


def laughter_instances(probs, thresholds=(0.5,), min_lengths=(0.2,), fps=100.0):
    probs = [fix_prob(float(p)) for p in probs]
    out = {}
    for thr in thresholds:
        for min_l in min_lengths:
            runs, cur = [], []
            for i, p in enumerate(probs):
                if p > thr:
                    cur.append(i)
                elif cur:
                    runs.append(cur)
                    cur = []
            if cur:
                runs.append(cur)
            inst = [(r[0] / fps, r[-1] / fps) for r in runs]
            out[(thr, min_l)] = [s for s in inst if s[1] - s[0] > min_l]
    return out


def run_indices(probs, thr):
    """Integer (first_frame, last_frame) runs for one threshold (the bit-exact part)."""
    probs = [fix_prob(float(p)) for p in probs]
    runs, start = [], None
    for i, p in enumerate(probs):
        if p > thr:
            if start is None:
                start = i
        elif start is not None:
            runs.append((start, i - 1))
            start = None
    if start is not None:
        runs.append((start, len(probs) - 1))
    return runs
