"""Child rank of tests/test_parallel_cpu.py::test_spawned_ranks_drive_the_real_sharding (started by parallel.spawn_ranks).

Drives the REAL data-parallel plumbing on CPU / gloo: parallel.init_from_env, load_data.load_segment_table (index
shuffle), load_data.SegmentSampler / SegmentLoader, train.run_epoch and parallel.GradReducer; only the model (the HIP
engine needs a GPU) and the feature gather are stand-ins."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "laughter-detection-icsi_amd")
for p in (os.path.join(PKG, "utils"), PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import load_data  # noqa: E402
import parallel  # noqa: E402
import train  # noqa: E402


class _Engine:
    def reset_optimizer(self):
        pass


class StubModel:
    """train_step = 'gradient' (one count per segment id) -> the real reducer -> log."""

    def __init__(self, n, reducer):
        self.engine, self.global_step, self.epoch, self.best_val_loss = _Engine(), 0, 0, float("inf")
        self.n, self.reducer = n, reducer
        self.seen_local, self.total, self.batch_labels = [], torch.zeros(n), []

    def train(self):
        pass

    def train_step(self, inputs, labels, max_norm=1.0, grad_reduce=None, grad_scale=1.0):
        ids = inputs[:, 0, 0].to(torch.int64)
        g = torch.zeros(self.n)
        g[ids] += 1.0
        grad_reduce(g)                  # dist.all_reduce over gloo: hangs (and the test times out) if a rank is missing
        self.total += g
        self.seen_local.extend(ids.tolist())
        self.batch_labels.append(sorted(set(labels.tolist())))
        self.global_step += 1
        return torch.tensor([0.5, 1.0, 1.0, 1.0, 1.0, float(len(ids)), 0.0, 0.0])


class StubDataset:
    def __init__(self, table):
        self.table = table

    def __getitem__(self, idx):
        idx = np.asarray(idx)
        x = torch.zeros(len(idx), 100, 44)
        x[:, 0, 0] = torch.from_numpy(idx.astype(np.float32))  # the segment's position in the epoch order
        return {"inputs": x, "is_laugh": torch.from_numpy(self.table.label[idx]), "input_lens": None, "cut": idx}


def main():
    out_dir, batch = sys.argv[1], int(sys.argv[2])
    rank, world, _ = parallel.init_from_env(backend="gloo")
    table = load_data.load_segment_table(out_dir, "train", world=world)  # <out_dir>/train_df.csv, written by the test
    sampler = load_data.SegmentSampler(len(table), max_cuts=batch, rank=rank, world=world, min_batch=2)
    loader = load_data.SegmentLoader(StubDataset(table), sampler)
    reducer = parallel.GradReducer()
    model = StubModel(len(table), reducer)
    for _ in range(2):  # two epochs: a rank one step short in epoch 1 would pair its collectives across epochs
        train.run_epoch(model, loader, None, out_dir, None, batch, [], reducer, rank=rank, verbose=False)
    res = {"rank": rank, "world": world, "steps": model.global_step, "len_sampler": len(sampler),
           "backend": reducer.backend, "calls": reducer.calls, "seen": model.seen_local,
           "total": model.total.tolist(), "batch_labels": model.batch_labels}
    json.dump(res, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps({"ok": True, "steps": model.global_step}))


if __name__ == "__main__":
    main()
