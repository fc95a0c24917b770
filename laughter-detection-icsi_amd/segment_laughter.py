#!/usr/bin/env python3
"""Sliding-window laughter segmentation of one audio file on the MI355X.

Counterpart of the reference's segment_laughter.py (argparse :28-40, model load :59-74, `load_and_pred` :79-122,
`save_instances` :124-161): same flags, same outputs (one TextGrid per (threshold, min_length) setting under
`<output_dir>/t_<thr>/l_<min_len>/`).  What changes is the loop: the reference moves 32 windows at a time through the
model (11,250 host round trips for a 60 min channel); here the whole file is featurised in one launch and
`engine.predict_windows` reads the stride-one-frame windows straight from the (T, 44) matrix, in chunks of
`engine.PREDICT_CHUNK[precision]` windows (the sizes bench.py measures).  `--precision fp32` (default) is the reference's
arithmetic; `--precision fp16` runs the convolutions on the 16-bit matrix cores (BASELINE configs[4]: about 30x faster;
tolerance in tests/test_resnet_gpu.py) -- the invocation behind the published real-time factor is

    python segment_laughter.py --config resnet_base --model_path <dir> --input_audio_file <wav> --output_dir <out> \\
        --precision fp16 --thresholds 0.1,...  --min_lengths 0.0,0.1,0.2

The script prints the real-time factor of everything it does (file read, featurisation, windows, threshold sweep,
TextGrid / wav output).  With torchrun (one process per GPU) the window range is sharded over ranks and the
probabilities are all-gathered.
"""
import argparse
import os
import sys
import time

_PKG = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(_PKG, "utils"), _PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import audio_utils  # noqa: E402
import config as config_mod  # noqa: E402
import laugh_segmenter  # noqa: E402
import load_data  # noqa: E402
import parallel  # noqa: E402
import textgrid  # noqa: E402
import torch_utils  # noqa: E402


def build_model(config_name, model_path, device):
    config = config_mod.MODEL_MAP[config_name]
    model = config['model'](dropout_rate=0.0, linear_layer_size=config['linear_layer_size'],
                            filter_sizes=config['filter_sizes'])
    model.set_device(device)
    ckpt = os.path.join(model_path, 'best.pth.tar')
    if os.path.exists(ckpt):
        torch_utils.load_checkpoint(ckpt, model, map_location=device)
        model.eval()
    else:
        raise Exception(f"Model checkpoint not found at {model_path}")
    return model


def predict_file(model, audio_path, chunk=None, rank=0, world=1, precision="fp32"):
    """probs (T,) float32 numpy for the stride-one-frame windows of the file + its duration in seconds.
    chunk=None: the engine's own chunk size for the precision (engine.PREDICT_CHUNK)."""
    loader = load_data.create_inference_dataloader(audio_path)
    feats = loader.dataset.feats
    T = feats.shape[0]
    sh = parallel.shard_indices(T, rank, world)
    local = model.engine.predict_windows(feats, chunk=chunk, start=sh.start, stop=sh.stop, precision=precision)
    probs = parallel.gather_probs(local, T, rank, world)
    file_length = audio_utils.get_audio_length(audio_path)  # seconds; fps = T / file_length (segment_laughter.py:103-104)
    return probs.cpu().numpy(), file_length


def save_audio_instances(instances, audio_path, output_dir):
    """One `laugh_<i>.wav` per instance (segment_laughter.py:133-149).  The reference re-reads the file with
    `librosa.load(sr=44100)`, i.e. resampled; here the cut is taken from the file's own samples at its own rate
    (no resampler on this path) -- same instants, same int16 scaling (`maxv = 32767`)."""
    from scipy.io import wavfile
    sr = audio_utils.get_sampling_rate(audio_path)
    y = load_data.load_audio(audio_path, sampling_rate=sr)
    maxv = np.iinfo(np.int16).max
    paths = []
    for index, instance in enumerate(instances):
        laughs = laugh_segmenter.cut_laughter_segments([instance], y, sr)
        wav_path = os.path.join(output_dir, "laugh_" + str(index) + ".wav")
        wavfile.write(wav_path, sr, (np.asarray(laughs, dtype=np.float64) * maxv).astype(np.int16))
        paths.append(wav_path)
    return paths


def load_and_pred(model, audio_path, thresholds, min_lengths, output_dir, save_to_textgrid=True, rank=0, world=1,
                  precision="fp32", save_to_audio_files=False, verbose=True, save_probs=None):
    """segment_laughter.py:79-122.  Returns (seconds taken by everything below, {(thr, min_len): [(start, end), ...]})."""
    if save_to_audio_files and output_dir is None:
        raise Exception("Need to specify an output directory to save audio files")   # segment_laughter.py:138-140
    start_time = time.time()
    probs, file_length = predict_file(model, audio_path, rank=rank, world=world, precision=precision)
    predict_time = time.time() - start_time
    if save_probs and rank == 0:
        np.save(save_probs, probs)   # (not in the reference: the per-frame track, e.g. to compare a sharded run with a single-rank one)
    fps = len(probs) / float(file_length)
    instance_dict = laugh_segmenter.get_laughter_instances(probs, thresholds=thresholds, min_lengths=min_lengths, fps=fps)
    sweep_time = time.time() - start_time - predict_time
    if rank == 0:
        for setting, instances in instance_dict.items():
            if verbose:
                print(f"Found {len(instances)} laughs for threshold {setting[0]} and min_length {setting[1]}.")
            out_dir = os.path.join(output_dir or '.', f't_{setting[0]}', f'l_{setting[1]}')
            if save_to_textgrid or (save_to_audio_files and len(instances) > 0):
                os.makedirs(out_dir, exist_ok=True)
            if save_to_audio_files and len(instances) > 0:
                wav_paths = save_audio_instances(instances, audio_path, out_dir)
                if verbose:
                    print(laugh_segmenter.format_outputs(instances, wav_paths))   # segment_laughter.py:148
            if save_to_textgrid:
                fname = os.path.splitext(os.path.basename(audio_path))[0]
                textgrid.write_laughter_textgrid(os.path.join(out_dir, fname + '.TextGrid'), instances, xmax=file_length)
    time_taken = time.time() - start_time
    if rank == 0:
        print(f'Completed in: {time_taken:.2f}s  (real-time factor of the whole script {time_taken / file_length:.2e} at '
              f'{precision}: read + featurise + {len(probs)} windows {predict_time:.3f}s, '
              f'{len(instance_dict)}-setting sweep {sweep_time:.3f}s, output {time_taken - predict_time - sweep_time:.3f}s)')
    return time_taken, instance_dict


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--model_path', type=str, default='checkpoints/in_use/resnet_with_augmentation')
    parser.add_argument('--config', type=str, default='resnet_base')
    parser.add_argument('--thresholds', type=str, default='0.5', help='Single value or comma-separated list of thresholds to evaluate')
    parser.add_argument('--min_lengths', type=str, default='0.2', help='Single value or comma-separated list of min_lengths to evaluate')
    parser.add_argument('--input_audio_file', required=True, type=str)
    parser.add_argument('--output_dir', type=str, default=None)
    parser.add_argument('--save_to_audio_files', type=str, default='False',
                        help="laugh_<i>.wav per instance, cut from the file at its own sampling rate (the reference resamples to 44.1 kHz "
                             "with librosa and defaults this flag to 'True'; here it is opt-in: it needs --output_dir)")
    parser.add_argument('--save_probs', type=str, default=None, help='(not in the reference) write the per-frame probabilities to this .npy')
    parser.add_argument('--save_to_textgrid', type=str, default='True')
    parser.add_argument('--gpus', type=int, default=None,
                        help='ranks that share the window range; > 1 without a launcher environment starts the ranks itself')
    parser.add_argument('--precision', type=str, default='fp32', choices=['fp32', 'fp16'], help='matrix-core precision')
    args = parser.parse_args(argv)
    if args.gpus is not None and args.gpus > 1 and not parallel.under_launcher():
        raise SystemExit(parallel.spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:] if argv is None else list(argv),
                                              timeout=None))
    thresholds = [float(t) for t in args.thresholds.split(',')]
    min_lengths = [float(l) for l in args.min_lengths.split(',')]
    rank, world, local = parallel.init_from_env()
    if args.gpus is not None and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher environment says WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("segment_laughter.py needs an MI355X (the HIP path has no CPU fallback)")
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    model = build_model(args.config, args.model_path, device)
    truthy = ('true', '1', 'yes')
    load_and_pred(model, args.input_audio_file, thresholds, min_lengths, args.output_dir,
                  save_to_textgrid=args.save_to_textgrid.lower() in truthy, rank=rank, world=world,
                  precision=args.precision, save_to_audio_files=args.save_to_audio_files.lower() in truthy, save_probs=args.save_probs)


if __name__ == '__main__':
    main()
