"""Minimal Praat TextGrid writer for the laughter tier (segment_laughter.py:150-161 writes one IntervalTier named
'laughter' with `tgt`; analysis/analyse.py:64-96 reads those files back).  Long ("ooTextFile") format, gaps filled
with empty intervals, which every TextGrid reader accepts."""


def write_laughter_textgrid(path, instances, xmax=None, tier_name="laughter", label="laugh"):
    inst = sorted((float(s), float(e)) for s, e in instances)
    end = max([xmax or 0.0] + [e for _, e in inst])
    intervals, t = [], 0.0
    for s, e in inst:
        if s > t:
            intervals.append((t, s, ""))
        intervals.append((s, e, label))
        t = e
    if end > t:
        intervals.append((t, end, ""))
    with open(path, "w") as f:
        f.write('File type = "ooTextFile"\nObject class = "TextGrid"\n\n')
        f.write(f"xmin = 0\nxmax = {end}\ntiers? <exists>\nsize = 1\nitem []:\n")
        f.write(f'    item [1]:\n        class = "IntervalTier"\n        name = "{tier_name}"\n')
        f.write(f"        xmin = 0\n        xmax = {end}\n        intervals: size = {len(intervals)}\n")
        for i, (s, e, txt) in enumerate(intervals, start=1):
            f.write(f"        intervals [{i}]:\n            xmin = {s}\n            xmax = {e}\n            text = \"{txt}\"\n")
