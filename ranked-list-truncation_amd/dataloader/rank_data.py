"""robust04-format ranked lists -> GPU batches.

On-disk contract (the reference's, dataloader/attncut_dataloader.py:21-59, choopy_dataloader.py:21-45):

    <base>/<retrieve_data>/<name>_{train,test}.pkl          dict[qid] -> dict[doc_id -> score], rank order
    <base>/<retrieve_data>/attncut/<name>_{train,test}.pkl  dict[qid] -> list[S][2] neighbour cosine features
    <base>/<retrieve_data>/gt.pkl                           dict[qid] -> list[doc_id] (relevant documents)

label[q][j] = 1 if the j-th ranked document of q is in gt[q] else 0.  AttnCut-family input is
(N,S,3) = [score | 2 features]; Choopy-family input is (N,S,1) = [score].

Unlike the reference (python lists of lists -> t.Tensor -> DataLoader workers) the set is packed once
into two contiguous pinned fp32 host arrays; an epoch is a permutation; every batch is one gather on the
host and one async H2D copy on a side stream, double-buffered against the compute stream.
"""
import os
import pickle

import numpy as np
import torch

DATASET_BASE = os.environ.get("RLT_DATASET_BASE", os.path.join(os.path.dirname(os.path.dirname(
    os.path.dirname(os.path.abspath(__file__)))), "dataset"))


def _load(path):
    with open(path, "rb") as f:
        return pickle.load(f)


def _pack(raw, stats, gt, with_stats):
    qids = list(raw.keys())
    n, s = len(qids), len(next(iter(raw.values())))
    feat = 3 if with_stats else 1
    x = np.empty((n, s, feat), dtype=np.float32)
    y = np.zeros((n, s), dtype=np.float32)
    for i, q in enumerate(qids):
        docs = raw[q]
        if len(docs) != s:
            raise ValueError(f"query {q}: {len(docs)} ranked documents, expected {s}")
        x[i, :, 0] = np.fromiter(docs.values(), dtype=np.float32, count=s)
        if with_stats:
            x[i, :, 1:] = np.asarray(stats[q], dtype=np.float32).reshape(s, 2)
        rel = gt.get(q, ())
        rel = rel if isinstance(rel, (set, frozenset)) else set(rel)
        y[i] = np.fromiter((1.0 if d in rel else 0.0 for d in docs.keys()), dtype=np.float32, count=s)
    return x, y, qids


class RankData:
    def __init__(self, retrieve_data="robust04", dataset_name="bm25", with_stats=True, base=None):
        base = os.path.join(base or DATASET_BASE, retrieve_data)
        gt = _load(os.path.join(base, "gt.pkl"))
        self.splits = {}
        for split in ("train", "test"):
            raw = _load(os.path.join(base, f"{dataset_name}_{split}.pkl"))
            stats = _load(os.path.join(base, "attncut", f"{dataset_name}_{split}.pkl")) if with_stats else None
            x, y, qids = _pack(raw, stats, gt, with_stats)
            xt, yt = torch.from_numpy(x), torch.from_numpy(y)
            if torch.cuda.is_available():
                xt, yt = xt.pin_memory(), yt.pin_memory()
            self.splits[split] = (xt, yt, qids)

    # the reference's accessor names (attncut_dataloader.py:61-71)
    def getX_train(self): return self.splits["train"][0]
    def getX_test(self): return self.splits["test"][0]
    def gety_train(self): return self.splits["train"][1]
    def gety_test(self): return self.splits["test"][1]


class BatchLoader:
    """Iterable over (X, y) batches; on a GPU the batches arrive already on the device
    (pinned gather + non_blocking copy on a side stream, one batch ahead)."""

    def __init__(self, x, y, batch_size, shuffle=True, device=None, seed=None, drop_last=False):
        self.x, self.y, self.batch_size, self.shuffle = x, y, batch_size, shuffle
        self.device = torch.device(device) if device is not None else None
        self.gen = torch.Generator().manual_seed(seed) if seed is not None else None
        self.drop_last = drop_last
        self._stream = torch.cuda.Stream(self.device) if (self.device is not None and self.device.type == "cuda") else None

    def __len__(self):
        n = self.x.shape[0]
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _host_batch(self, idx):
        xb, yb = self.x.index_select(0, idx), self.y.index_select(0, idx)
        if self._stream is not None:
            xb, yb = xb.pin_memory(), yb.pin_memory()
        return xb, yb

    def _to_device(self, xb, yb):
        if self._stream is None:
            return (xb, yb) if self.device is None else (xb.to(self.device), yb.to(self.device))
        with torch.cuda.stream(self._stream):
            xd, yd = xb.to(self.device, non_blocking=True), yb.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        return xd, yd, ev, (xb, yb)

    def __iter__(self):
        n = self.x.shape[0]
        order = torch.randperm(n, generator=self.gen) if self.shuffle else torch.arange(n)
        chunks = [order[i:i + self.batch_size] for i in range(0, n, self.batch_size)]
        if self.drop_last and chunks and len(chunks[-1]) < self.batch_size:
            chunks.pop()
        if self._stream is None:
            for idx in chunks:
                yield self._to_device(*self._host_batch(idx))
            return
        pending = self._to_device(*self._host_batch(chunks[0])) if chunks else None
        for i in range(len(chunks)):
            xd, yd, ev, _keep = pending
            pending = self._to_device(*self._host_batch(chunks[i + 1])) if i + 1 < len(chunks) else None
            torch.cuda.current_stream(self.device).wait_event(ev)
            yield xd, yd


def _loaders(rank_data, batch_size, device, seed):
    xtr, ytr, _ = rank_data.splits["train"]
    xte, yte, _ = rank_data.splits["test"]
    # the reference shuffles BOTH loaders (attncut_dataloader.py:87-88)
    return (BatchLoader(xtr, ytr, batch_size, True, device, seed),
            BatchLoader(xte, yte, batch_size, True, device, None if seed is None else seed + 1), rank_data)


def attncut_dataloader(retrieve_data="robust04", dataset_name="bm25", batch_size=20, device=None, base=None, seed=None):
    return _loaders(RankData(retrieve_data, dataset_name, True, base), batch_size, device, seed)


def choopy_dataloader(retrieve_data="robust04", dataset_name="bm25", batch_size=20, device=None, base=None, seed=None):
    return _loaders(RankData(retrieve_data, dataset_name, False, base), batch_size, device, seed)
