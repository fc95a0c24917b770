"""CPU tests of the host-side integer logic: segmenter vs reference goldens, segment table vs the oracle restatement
and the reference's own sample tables, InferenceDataset window semantics."""
import csv
import json
import os

import numpy as np
import pytest

from oracle import recipe, segmenter_oracle as so, segments_oracle as sgo


def test_segmenter_matches_reference_golden(golden_dir):
    import laugh_segmenter as ls
    cases = json.load(open(os.path.join(golden_dir, "segmenter.json")))
    assert len(cases) >= 3
    for c in cases:
        p = recipe.make_prob_track(c["seed"], c["n"]) if "seed" in c else np.array(c["probs"])
        d = ls.get_laughter_instances(p, c["thresholds"], c["min_lengths"], c["fps"])
        ref = {tuple(k): [tuple(s) for s in v] for k, v in c["result"]}
        assert list(d.keys()) == list(ref.keys())
        for k in ref:
            assert d[k] == ref[k], (c.get("seed", c.get("name")), k)


@pytest.mark.parametrize("n,seed", [(1, 0), (2, 1), (1000, 2), (36000, 3)])
def test_segmenter_run_indices_bit_exact_vs_oracle(n, seed):
    import laugh_segmenter as ls
    p = recipe.make_prob_track(seed, n) if n >= 100 else np.random.default_rng(seed).random(n)
    for thr in (0.0, 0.3, 0.5, 0.99, 1.0):
        got = [tuple(int(v) for v in r) for r in ls.get_laughter_frame_spans(p, thr)]
        assert got == so.run_indices(p, thr)
    # float32 probabilities straight from the GPU head behave the same
    p32 = p.astype(np.float32)
    assert [tuple(int(v) for v in r) for r in ls.get_laughter_frame_spans(p32, 0.5)] == so.run_indices(p32, 0.5)


def test_segmenter_edge_cases():
    import laugh_segmenter as ls
    assert ls.get_laughter_instances([], [0.5], [0.2], 100.0) == {(0.5, 0.2): []}
    d = ls.get_laughter_instances([0.9] * 30, [0.5], [0.2, 0.3], 100.0)
    assert d[(0.5, 0.2)] == [(0.0, 0.29)] and d[(0.5, 0.3)] == []  # 0.29 - 0.0 > 0.2 only
    d = ls.get_laughter_instances([0.0, -1.0, 2.0, 0.0], [0.0], [0.0], 1.0)
    assert d[(0.0, 0.0)] == [(0.0, 3.0)]  # p <= 0 becomes 1e-7 > 0; p > 1 becomes 1
    assert ls.fix_over_underflow(1.5) == 1 and ls.fix_over_underflow(0) == 0.0000001 and ls.fix_over_underflow(0.3) == 0.3


@pytest.mark.parametrize("name", ["sample_df.csv", "tiny_sample.csv"])
def test_segment_table_matches_oracle_on_reference_sample_tables(golden_dir, name):
    import segments
    path = os.path.join(golden_dir, "data_dfs", name)
    rows = list(csv.DictReader(open(path)))
    assert list(rows[0].keys()) == ["start", "duration", "sub_start", "sub_duration", "audio_path", "meeting_id", "chan_id", "label"]
    t = segments.table_from_csv(path)
    chans, ref = sgo.rows_to_segments(rows)
    assert t.channels == chans and len(t) == len(ref)
    got = list(zip(t.channel.tolist(), t.first_frame.tolist(), t.n_frames.tolist(), t.label.tolist()))
    assert got == ref
    assert t.n_frames.max() <= 100 and t.first_frame.dtype == np.int64 and t.label.dtype == np.int32
    # first row of the sample table: sub_start 1478.92 s -> frame 147892, one full second
    assert got[0][1:3] == (147892, 100)


def test_segment_table_short_ragged_and_bad_rows():
    import segments
    rows = [dict(start=0, duration=0.5, sub_start=0.0, sub_duration=0.37, audio_path="a.wav", meeting_id="m", chan_id="c", label=1),
            dict(start=3, duration=2, sub_start=3.29, sub_duration=1.0, audio_path="b.wav", meeting_id="m", chan_id="d", label=0)]
    t = segments.table_from_rows(rows)
    assert t.n_frames.tolist() == [37, 100] and t.first_frame.tolist() == [0, 329]  # 3.29 / 0.01 = 328.99999999999994 -> 329
    assert segments.table_from_rows([]).label.shape == (0,)
    with pytest.raises(ValueError):
        segments.table_from_rows([dict(rows[0], label=2)])
    with pytest.raises(ValueError):
        segments.table_from_rows([dict(rows[0], sub_start=-1.0)])
    sh = t.shuffled(3)
    assert sorted(sh.first_frame.tolist()) == [0, 329]
    assert len(t.shard(0, 2)) == 1 and len(t.shard(1, 2)) == 1 and len(t.shard(2, 3)) == 0


def test_whole_track_windows_and_labels():
    import segments
    laughs = [(1500, 1800), (3000, 3001), (5999, 7000)]
    t = segments.whole_track_table(799, "Bmr021/chan3.sph", laughs)
    ref = sgo.whole_track_windows(799, laughs)
    assert len(t) == 7 == len(ref)  # the 8th window is partial and dropped
    assert list(zip(t.first_frame.tolist(), t.n_frames.tolist(), t.label.tolist())) == ref
    # (1000,2000] overlaps (1500,1800]; (3000,4000] overlaps (3000,3001]; (2000,3000] does NOT touch (3000,3001]
    assert t.label.tolist() == [0, 1, 0, 1, 0, 1, 1]


def test_inference_dataset_window_semantics():
    import datasets
    feats = np.arange(250 * 44, dtype=np.float32).reshape(250, 44)
    ds = datasets.InferenceDataset(feats)
    assert len(ds) == 250
    assert np.array_equal(ds[0], feats[:100])
    last = ds[249]
    assert last.shape == (100, 44) and np.array_equal(last[0], feats[249]) and np.all(last[1:] == 0.0)
    mid = ds[200]
    assert np.array_equal(mid[:50], feats[200:]) and np.all(mid[50:] == 0.0)


def test_parallel_window_shards_cover_track():
    import parallel
    T = 360000
    seen = 0
    for r in range(8):
        sh = parallel.shard_indices(T, r, 8)
        assert sh.start == seen
        seen = sh.stop
    assert seen == T


def test_get_audio_length_reads_header_only(tmp_path):
    """audio_utils.get_audio_length (reference audio_utils.py:7-9): samples / rate, also for a ragged sample count."""
    import audio_utils
    import load_data
    from scipy.io import wavfile
    n = 16000 * 2 + 37
    x = (np.sin(np.arange(n) * 0.01) * 0.3).astype(np.float32)
    wavfile.write(str(tmp_path / "a.wav"), 16000, (x * 32767).astype(np.int16))
    wavfile.write(str(tmp_path / "f.wav"), 16000, x)  # IEEE-float wav: the `wave` module rejects it
    np.save(str(tmp_path / "a.npy"), x)
    for name in ("a.wav", "f.wav", "a.npy"):
        p = str(tmp_path / name)
        assert audio_utils.get_audio_length(p) == n / 16000.0
        assert load_data.load_audio(p).shape == (n,)
    with pytest.raises(ValueError):
        audio_utils.get_audio_length(str(tmp_path / "a.sph"))


def test_rank_manifests_of_a_data_parallel_featurisation_are_found_and_foreign_features_refused(tmp_path):
    """compute_features.py under N ranks leaves {split}_feats.rank<r>.jsonl (no {split}_feats.jsonl): train.find_feats_manifests
    must find them (it silently re-featurised from audio), load_data.read_feats_manifest gathers them, and a record
    written by ANOTHER extractor configuration (frame shift, filter bank, ...) is refused instead of being trained on."""
    import argparse
    import json
    import types

    import load_data
    import train
    cut = tmp_path / "lhotse" / "cutsets"
    cut.mkdir(parents=True)
    cfg_now = {"sampling_rate": 16000, "frame_shift": 0.01, "num_filters": 44, "mel_variant": "kaldi"}
    ext = types.SimpleNamespace(config=types.SimpleNamespace(to_dict=lambda: dict(cfg_now, device="cuda")))
    assert load_data.extractor_signature(ext) == cfg_now   # the device does not decide the numbers
    feats = np.arange(300 * 44, dtype=np.float32).reshape(300, 44)
    for r, name in enumerate(("a", "b")):
        np.save(tmp_path / f"{name}.npy", feats + r)
        rec = {"id": name, "audio_path": str(tmp_path / f"{name}.wav"), "features_path": str(tmp_path / f"{name}.npy"),
               "num_frames": 300, "num_features": 44, "frame_shift": 0.01, "extractor": cfg_now}
        (cut / f"train_feats.rank{r}.jsonl").write_text(json.dumps(rec) + "\n")
    args = argparse.Namespace(feats_manifest_dir=None, data_root=str(tmp_path), lhotse_dir="lhotse")
    found = train.find_feats_manifests(args)
    assert found == {"train": str(cut / "train_feats.jsonl")}            # the stem stands for the rank files
    assert train.find_feats_manifests(argparse.Namespace(feats_manifest_dir=str(cut), data_root="x", lhotse_dir="y")) == found
    with pytest.raises(SystemExit):
        train.find_feats_manifests(argparse.Namespace(feats_manifest_dir=str(tmp_path), data_root="x", lhotse_dir="y"))
    man = load_data.read_feats_manifest(found["train"])
    got = load_data._stored_features(man, str(tmp_path / "b.wav"), 44, ext)
    assert np.array_equal(got, feats + 1)
    assert load_data._stored_features(man, str(tmp_path / "nobody.wav"), 44, ext) is None
    other = types.SimpleNamespace(config=types.SimpleNamespace(to_dict=lambda: dict(cfg_now, frame_shift=0.0125, device="cuda")))
    with pytest.raises(ValueError, match="another extractor configuration.*frame_shift"):
        load_data._stored_features(man, str(tmp_path / "a.wav"), 44, other)
    # a manifest of round 4 (no extractor record) is still checked for what it does hold: the frame shift
    old = {k: v for k, v in json.loads((cut / "train_feats.rank0.jsonl").read_text()).items() if k != "extractor"}
    (cut / "train_feats.rank0.jsonl").write_text(json.dumps(old) + "\n")
    man = load_data.read_feats_manifest(found["train"])
    assert load_data._stored_features(man, str(tmp_path / "a.wav"), 44, ext) is not None
    with pytest.raises(ValueError, match="frame shift"):
        load_data._stored_features(man, str(tmp_path / "a.wav"), 44, other)


def _read_laughter_textgrid(path):
    """What analysis/analyse.py:39-46 takes from a file: `grid['laughter']` of praat-textgrids, every interval's xmin / xmax / text
    (long "ooTextFile" format: `item [k]:` -> `name`, `intervals [i]:` -> `xmin`, `xmax`, `text`), kept iff text == 'laugh'."""
    import re
    text = open(path).read()
    assert text.startswith('File type = "ooTextFile"\nObject class = "TextGrid"\n')
    head, *items = re.split(r"\n\s*item \[\d+\]:\n", text)
    tiers = {}
    for item in items:
        assert re.search(r'class = "IntervalTier"', item)
        name = re.search(r'name = "([^"]*)"', item).group(1)
        n = int(re.search(r"intervals: size = (\d+)", item).group(1))
        ivs = [(float(a), float(b), t) for a, b, t in
               re.findall(r'intervals \[\d+\]:\s*xmin = (\S+)\s*xmax = (\S+)\s*text = "([^"]*)"', item)]
        assert len(ivs) == n
        tiers[name] = ivs
    xmax = float(re.search(r"\nxmax = (\S+)", head).group(1))
    assert int(re.search(r"\nsize = (\d+)", head).group(1)) == len(tiers)
    grid = tiers["laughter"]
    # a tier praat accepts: contiguous from 0 to xmax, no empty or reversed interval
    t = 0.0
    for a, b, _ in grid:
        assert a == t and b > a
        t = b
    assert t == xmax or not grid
    return [(a, b) for a, b, txt in grid if str(txt) == "laugh"], xmax


def test_textgrid_round_trip_gives_back_the_instances(tmp_path):
    """The TextGrid the path writes (segment_laughter.py:150-161) read back the way the reference's evaluation reads it
    (analysis/analyse.py:39-46): exactly the instances of get_laughter_instances, bit for bit."""
    import laugh_segmenter as ls
    import textgrid
    p = recipe.make_prob_track(5, 3000)
    p[-40:] = 0.99                                       # a laugh that runs to the last frame of the file
    fps, file_length = 100.0, 30.0
    d = ls.get_laughter_instances(p, [0.2, 0.5, 0.8], [0.0, 0.2], fps)
    assert any(v and v[-1][1] == (len(p) - 1) / fps for v in d.values())
    cases = list(d.items()) + [(("empty", 0), []), (("to_xmax", 0), [(1.25, 2.5), (29.0, file_length)]),
                               (("from_zero", 0), [(0.0, 0.31)]), (("touching", 0), [(1.0, 2.0), (2.0, 3.5)])]
    for key, instances in cases:
        path = tmp_path / f"{key[0]}_{key[1]}.TextGrid"
        textgrid.write_laughter_textgrid(str(path), instances, xmax=file_length)
        got, xmax = _read_laughter_textgrid(str(path))
        assert got == [(float(a), float(b)) for a, b in instances], key
        assert xmax == file_length
    # no file length given: the tier ends at the last instance; nothing at all: an empty tier
    textgrid.write_laughter_textgrid(str(tmp_path / "a.TextGrid"), [(0.5, 0.75)])
    assert _read_laughter_textgrid(str(tmp_path / "a.TextGrid")) == ([(0.5, 0.75)], 0.75)
    textgrid.write_laughter_textgrid(str(tmp_path / "b.TextGrid"), [])
    assert _read_laughter_textgrid(str(tmp_path / "b.TextGrid")) == ([], 0.0)


def test_format_outputs_is_the_references():
    """laugh_segmenter.py:141-149."""
    import laugh_segmenter as ls
    inst = [(0.5, 1.25), (3.0, 3.5)]
    assert ls.format_outputs(inst) == [{'start': 0.5, 'end': 1.25}, {'start': 3.0, 'end': 3.5}]
    out = ls.format_outputs(inst, ["a.wav", "b.wav"])
    assert out == [{'filename': "a.wav", 'start': 0.5, 'end': 1.25}, {'filename': "b.wav", 'start': 3.0, 'end': 3.5}]
    assert list(out[0]) == ['filename', 'start', 'end']
    assert ls.format_outputs([]) == []
    with pytest.raises(IndexError):
        ls.format_outputs(inst, ["a.wav"])


def test_librosa_convention_host_tables_are_the_third_party_goldens(golden_dir):
    """The product's librosa-convention tables (feats.slaney_mel_bank, hann_window_periodic, dct2_ortho: the only place the
    convention enters -- the kernel is the same) against tests/golden/librosa_conv.npz (transformers.audio_utils) and
    scipy.fft.dct.  Host arithmetic only: nothing here needs the GPU."""
    import feats
    g = np.load(os.path.join(golden_dir, "librosa_conv.npz"))
    for n_mels in (44, 40, 128):
        assert np.abs(feats.slaney_mel_bank(n_mels, 16000) - g[f"bank_{n_mels}"]).max() < 1e-15
    scipy_fft = pytest.importorskip("scipy.fft")
    eye = np.eye(44)
    assert np.abs(feats.dct2_ortho(44, 20) - scipy_fft.dct(eye, type=2, norm="ortho", axis=0)[:20].T).max() < 1e-14
    sig = pytest.importorskip("scipy.signal")
    assert np.abs(feats.hann_window_periodic(400) - sig.get_window("hann", 400, fftbins=True)).max() < 1e-15
