"""Child rank of tests/test_parallel_cpu.py::test_data_parallel_logging_equals_single_process (parallel.spawn_ranks, gloo).

Drives train.run_epoch with logging + validation through the real sampler / loader / reducer; the model is a stand-in whose
counters are a deterministic function of the segments it is given, so the logged metrics of a 2-rank run can be compared
with those of one process over the same global batches."""
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


def stub_probs(ids):
    return ((ids.to(torch.int64) * 37) % 101).to(torch.float32) / 100.0


def stub_counters(ids, labels):
    """[mean 'loss', #correct, #pred+, #true+, #target+, n, 0, 0] -- the head's layout (engine.metrics_from_counters)."""
    p = stub_probs(ids)
    pred = (p > 0.5)
    lab = labels.to(torch.bool)
    return torch.tensor([float(p.mean()), float((pred == lab).sum()), float(pred.sum()), float((pred & lab).sum()),
                         float(lab.sum()), float(len(ids)), 0.0, 0.0])


class _Engine:
    def reset_optimizer(self):
        pass

    def eval_metrics(self, probs, labels):
        return stub_counters(probs, labels)       # `probs` carries the segment ids (StubModel.predict)


class StubModel:
    def __init__(self, reducer):
        self.engine, self.global_step, self.epoch, self.best_val_loss = _Engine(), 0, 0, float("inf")
        self.reducer, self.predict_calls, self.accum_seen = reducer, 0, []

    def train(self):
        pass

    def eval(self):
        pass

    def state_dict(self):
        return {}

    def predict(self, inputs):
        self.predict_calls += 1
        return inputs[:, 0, 0]

    def train_step(self, inputs, labels, max_norm=1.0, grad_reduce=None, grad_scale=1.0, grad_accum=1):
        ids = inputs[:, 0, 0]
        grad_reduce(torch.zeros(4))
        self.accum_seen.append(grad_accum)
        self.global_step += 1
        return stub_counters(ids, labels)


class StubDataset:
    def __init__(self, table):
        self.table = table

    def __getitem__(self, idx):
        idx = np.asarray(idx)
        x = torch.zeros(len(idx), 100, 44)
        x[:, 0, 0] = torch.from_numpy(idx.astype(np.float32))
        return {"inputs": x, "is_laugh": torch.from_numpy(self.table.label[idx]), "input_lens": None, "cut": idx}


def main():
    out_dir, batch, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rank, world, _ = parallel.init_from_env(backend="gloo")
    table = load_data.load_segment_table(out_dir, "train", world=world)
    dev = load_data.load_segment_table(out_dir, "dev", shuffle=True, seed=5)
    loader = load_data.SegmentLoader(StubDataset(table), load_data.SegmentSampler(len(table), max_cuts=batch, rank=rank, world=world, min_batch=2))
    val_loader = load_data.SegmentLoader(StubDataset(dev), load_data.SegmentSampler(len(dev), max_cuts=4))
    reducer = parallel.GradReducer()
    model = StubModel(reducer)
    rows = []
    ckpt = os.path.join(out_dir, f"ckpt_{tag}_{rank}")   # (per rank here, to see who writes)
    loss_sum = train.run_epoch(model, loader, val_loader, ckpt, 2, batch, rows, reducer, rank=rank, verbose=False)
    res = {"rank": rank, "world": world, "rows": rows, "loss_sum": loss_sum, "predict_calls": model.predict_calls,
           "steps": model.global_step, "wrote_checkpoint": os.path.exists(os.path.join(ckpt, "last.pth.tar"))}
    json.dump(res, open(os.path.join(out_dir, f"{tag}_rank{rank}.json"), "w"))
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
