#!/usr/bin/env python3
"""Build tests/golden/demo_clips.npz and tests/golden/demo_fbank_plot.npz from the reference's Demo.ipynb.

TEST INFRASTRUCTURE, BUILD CONTAINER ONLY (reads /root/reference, needs PIL + matplotlib for the PNG / colour map):

    python oracle/make_demo_fbank_golden.py [--reference /root/reference]

Demo.ipynb is the only place where the reference holds OUTPUT of its Lhotse feature extractor:

  cell 5   extractor = Fbank(FbankConfig(num_filters=40, frame_shift=1/100))
  cell 7   laugh_cut     = MonoCut(start=291.39,  duration=1.0, recording Bmr021_c0).compute_and_store_features(extractor, Lilcom)
           .plot_audio()  -> PNG of the waveform      .plot_features() -> PNG of the feature matrix      .play_audio() -> wav
  cell 9   non_laugh_cut = MonoCut(start=1478.92, duration=1.0, same recording): the same three outputs

What is extracted (data, never source):
  * the two embedded wav players -> 16,000 int16 samples each.  IPython's Audio() peak-normalises, so these are the cut's
    samples times 32767/max|x|.  The ORIGINAL integer samples are recovered exactly: they are the unique scale M for
    which x_norm * M / 32767 is integral (M = max|x_orig| in int16 units: 4967 and 7; cross-checked by eye against the
    y axes of the two plot_audio() images, -0.152 = -4967/32768 and -0.00021 = -7/32768).
  * the two plot_features() images: lhotse draws plt.matshow(np.flip(features.T, 0)): 40 rows (filter 39 on top) x 100
    columns (frames), matplotlib's default normalisation (min -> 0, max -> 1) and colour map (viridis).  Every cell's
    centre pixel is mapped back through the 256-entry viridis table (max squared RGB distance 3: the inversion is exact) ->
    uint8 level 0..255 per cell.

So the fixture gives, for two real recordings, the reference's own 40-filter log-mel features after lilcom (lossy,
tick 2^-5) and an 8-bit colour quantisation of their [min, max] range: one level = 0.059 (clip 0) / 0.019 (clip 1) in the
natural-log domain.  tests/test_oracle_golden.py compares oracle/fbank_oracle.py with it for both mel-bank candidates.
"""
import argparse
import base64
import io
import json
import os
import re
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def wav_of_cell(cell):
    for o in cell["outputs"]:
        html = "".join(o.get("data", {}).get("text/html", []))
        m = re.search(r"data:audio/wav;base64,([A-Za-z0-9+/=]+)", html)
        if m:
            with wave.open(io.BytesIO(base64.b64decode(m.group(1)))) as w:
                assert w.getnchannels() == 1 and w.getsampwidth() == 2
                return w.getframerate(), np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").copy()
    raise ValueError("no embedded wav")


def pngs_of_cell(cell):
    from PIL import Image
    return [np.asarray(Image.open(io.BytesIO(base64.b64decode(o["data"]["image/png"]))).convert("RGB")).astype(int)
            for o in cell["outputs"] if "image/png" in o.get("data", {})]


def original_scale(x_norm):
    """Smallest M with x_norm * M / 32767 integral up to the player's truncation (|error| <= M / 32767 < 0.2)."""
    x = x_norm.astype(np.float64)
    for m in range(1, 32768):
        y = x * m / 32767.0
        if np.abs(y - np.round(y)).max() <= m / 32767.0 + 1e-9 and np.abs(np.round(y)).max() == m:
            return m
    raise ValueError("no integral scale")


def decode_matshow(rgb, rows=40, cols=100):
    import matplotlib
    lut = (matplotlib.colormaps["viridis"](np.linspace(0, 1, 256))[:, :3] * 255).round().astype(int)
    coloured = (np.abs(rgb[..., 0] - rgb[..., 1]) + np.abs(rgb[..., 1] - rgb[..., 2])) > 30
    ys, xs = np.where(coloured)
    y0, y1, x0, x1 = ys.min(), ys.max() + 1, xs.min(), xs.max() + 1
    levels = np.zeros((rows, cols), np.uint8)
    worst = 0
    for r in range(rows):
        for c in range(cols):
            px = rgb[int(y0 + (r + 0.5) * (y1 - y0) / rows), int(x0 + (c + 0.5) * (x1 - x0) / cols)]
            d = ((lut - px) ** 2).sum(1)
            levels[r, c] = d.argmin()
            worst = max(worst, int(d.min()))
    return levels, worst, (int(y0), int(y1), int(x0), int(x1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    nb = json.load(open(os.path.join(args.reference, "Demo.ipynb")))
    clips, orig, scales, plots = [], [], [], []
    for ci in (7, 9):
        cell = nb["cells"][ci]
        assert "plot_features()" in "".join(cell["source"])
        sr, x = wav_of_cell(cell)
        assert sr == 16000 and x.shape == (16000,)
        m = original_scale(x)
        clips.append(x)
        scales.append(m)
        orig.append(np.round(x.astype(np.float64) * m / 32767.0).astype(np.int16))
        images = pngs_of_cell(cell)          # [plot_audio, plot_features]
        levels, worst, box = decode_matshow(images[1])
        assert worst <= 3, worst
        plots.append(levels)
        print(f"cell {ci}: wav peak-normalised from max|x| = {m} (int16 units = {m / 32768:.6f}); plot box {box}, "
              f"{len(np.unique(levels))} colour levels used, LUT inversion residual {worst}")
    np.savez_compressed(os.path.join(GOLDEN, "demo_clips.npz"), sr=np.array([16000, 16000]), clip0=clips[0], clip1=clips[1])
    np.savez_compressed(os.path.join(GOLDEN, "demo_fbank_plot.npz"), levels0=plots[0], levels1=plots[1],
                        orig0=orig[0], orig1=orig[1], scale=np.array(scales), num_filters=np.array(40))


if __name__ == "__main__":
    main()
