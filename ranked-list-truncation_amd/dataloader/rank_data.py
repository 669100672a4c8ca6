"""robust04-format ranked lists -> GPU batches.

On-disk contract (the reference's, dataloader/attncut_dataloader.py:21-59, choopy_dataloader.py:21-45,
mtcut_dataloader.py:21-59):

    <base>/<retrieve_data>/<name>_{train,test}.pkl          dict[qid] -> dict[doc_id -> score], rank order
    <base>/<retrieve_data>/attncut/<name>_{train,test}.pkl  dict[qid] -> list[S][W] per-document statistics
                                                            (robust04: W = 2 neighbour cosine features)
    <base>/<retrieve_data>/mtcut/<name>_{train,test}.pkl    same layout, the MQ2007 multi-task statistics (W = 46)
    <base>/<retrieve_data>/gt.pkl                           dict[qid] -> list[doc_id] (relevant documents)

label[q][j] = 1 if the j-th ranked document of q is in gt[q] else 0.  AttnCut-family input is
(N,S,1+W) = column_stack(score, statistics); Choopy-family input is (N,S,1) = [score].

Unlike the reference (python lists of lists -> t.Tensor -> DataLoader workers) a split is packed once
into contiguous pinned fp32 host arrays, ONE PAIR PER LIST LENGTH: the reference's files hold 300
documents for every query, but BASELINE configs[4] mixes lengths (k in {100, 200, 300}), and the models
take one length per batch (no padding or masking exists anywhere in the reference),
so lists are bucketed by length and every batch is homogeneous.  An epoch is one permutation per bucket;
the batches of the buckets are served round-robin; every batch is one gather into one of two reused
pinned staging buffers and one async H2D copy on a side stream, one batch ahead of the compute stream.
Under torch.distributed every rank draws the SAME permutations (one seed broadcast from rank 0), so
that rank r's shard of a batch (rlt_hip.parallel.shard_bounds) is a true partition of that batch.
"""
import os
import pickle

import numpy as np
import torch
import torch.distributed as dist

DATASET_BASE = os.environ.get("RLT_DATASET_BASE", os.path.join(os.path.dirname(os.path.dirname(
    os.path.dirname(os.path.abspath(__file__)))), "dataset"))


def _load(path):
    with open(path, "rb") as f:
        return pickle.load(f)


def _pack(raw, stats, gt):
    """dict-of-dicts -> {length: (x (n,length,F) f32, y (n,length) f32, qids)}; F = 1 + statistics width."""
    by_len = {}
    for q, docs in raw.items():
        by_len.setdefault(len(docs), []).append(q)
    out = {}
    for s, qids in by_len.items():
        width = 0 if stats is None else int(np.asarray(stats[qids[0]], dtype=np.float32).reshape(s, -1).shape[1])
        x = np.empty((len(qids), s, 1 + width), dtype=np.float32)
        y = np.zeros((len(qids), s), dtype=np.float32)
        for i, q in enumerate(qids):
            docs = raw[q]
            x[i, :, 0] = np.fromiter(docs.values(), dtype=np.float32, count=s)
            if width:
                st = np.asarray(stats[q], dtype=np.float32)
                if st.size != s * width:
                    raise ValueError(f"query {q}: statistics of shape {st.shape}, expected ({s}, {width})")
                x[i, :, 1:] = st.reshape(s, width)
            rel = gt.get(q, ())
            rel = rel if isinstance(rel, (set, frozenset)) else set(rel)
            y[i] = np.fromiter((1.0 if d in rel else 0.0 for d in docs.keys()), dtype=np.float32, count=s)
        out[s] = (x, y, qids)
    return out


def _pin(t):
    return t.pin_memory() if torch.cuda.is_available() else t


class RankData:
    """`stats_dir`: None (Choopy family: score only), 'attncut' or 'mtcut'."""

    def __init__(self, retrieve_data="robust04", dataset_name="bm25", with_stats=True, base=None, stats_dir="attncut"):
        base = os.path.join(base or DATASET_BASE, retrieve_data)
        gt = _load(os.path.join(base, "gt.pkl"))
        self.buckets = {}
        for split in ("train", "test"):
            raw = _load(os.path.join(base, f"{dataset_name}_{split}.pkl"))
            stats = _load(os.path.join(base, stats_dir, f"{dataset_name}_{split}.pkl")) if with_stats else None
            self.buckets[split] = {s: (_pin(torch.from_numpy(x)), _pin(torch.from_numpy(y)), qids)
                                   for s, (x, y, qids) in sorted(_pack(raw, stats, gt).items())}
        # one feature width for every bucket of both splits (the statistics width is inferred per bucket from its first
        # query): a model is built for ONE input size, so a mismatch is a data error - say so at load time
        widths = {(split, s): b[0].shape[2] for split, bs in self.buckets.items() for s, b in bs.items()}
        if len(set(widths.values())) > 1:
            raise ValueError(f"feature width differs between length buckets / splits: {widths}")

    @property
    def lengths(self):
        """List lengths of the TRAIN split (what the model is built for); `test_lengths` for the test split."""
        return sorted(self.buckets["train"])

    @property
    def test_lengths(self):
        return sorted(self.buckets["test"])

    @property
    def n_features(self):
        return next(iter(self.buckets["train"].values()))[0].shape[2]

    def _only(self, split, i):
        b = self.buckets[split]
        if len(b) != 1:
            raise ValueError(f"the {split} split holds lists of lengths {sorted(b)}; use .buckets[{split!r}][length]")
        return next(iter(b.values()))[i]

    # the reference's accessor names (attncut_dataloader.py:61-71); defined for single-length sets, as the reference's are
    def getX_train(self): return self._only("train", 0)
    def getX_test(self): return self._only("test", 0)
    def gety_train(self): return self._only("train", 1)
    def gety_test(self): return self._only("test", 1)


def shared_seed(seed=None):
    """A loader seed every rank agrees on: the given one, else a fresh draw on rank 0 broadcast to all ranks (without
    this each rank would draw its own permutation and the per-rank shards would overlap / miss lists)."""
    if seed is None:
        seed = int(torch.empty((), dtype=torch.int64).random_(0, 2 ** 62).item())
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            box = [seed]
            dist.broadcast_object_list(box, src=0)
            seed = int(box[0])
    return seed


class BatchLoader:
    """Iterable over homogeneous (X, y) batches of one or several length buckets; on a GPU the batches arrive already on
    the device (gather into a reused pinned staging buffer + non_blocking copy on a side stream, one batch ahead).

    buckets: list of (x (n,S,F), y (n,S)) pairs, one per list length."""

    def __init__(self, buckets, batch_size, shuffle=True, device=None, seed=None, drop_last=False):
        if isinstance(buckets, tuple) and len(buckets) == 2 and torch.is_tensor(buckets[0]):
            buckets = [buckets]
        self.buckets = [(x, y) for x, y in buckets]
        self.batch_size, self.shuffle, self.drop_last = batch_size, shuffle, drop_last
        self.device = torch.device(device) if device is not None else None
        self.gen = torch.Generator().manual_seed(shared_seed(seed))
        self._cuda = self.device is not None and self.device.type == "cuda"
        self._stream = torch.cuda.Stream(self.device) if self._cuda else None
        self._staging = {}            # (bucket, slot) -> (pinned x, pinned y, copy-done event)

    def __len__(self):
        bs = self.batch_size
        return sum(x.shape[0] // bs if self.drop_last else (x.shape[0] + bs - 1) // bs for x, _ in self.buckets)

    def _schedule(self):
        """[(bucket, index tensor)]: per-bucket permutation cut into batches, buckets interleaved round-robin."""
        per = []
        for x, _ in self.buckets:
            n = x.shape[0]
            order = torch.randperm(n, generator=self.gen) if self.shuffle else torch.arange(n)
            chunks = [order[i:i + self.batch_size] for i in range(0, n, self.batch_size)]
            if self.drop_last and chunks and len(chunks[-1]) < self.batch_size:
                chunks.pop()
            per.append(chunks)
        out = []
        for i in range(max((len(c) for c in per), default=0)):
            out.extend((b, c[i]) for b, c in enumerate(per) if i < len(c))
        return out

    def _stage(self, b, idx, slot):
        x, y = self.buckets[b]
        n = idx.numel()
        if not self._cuda:
            xb, yb = x.index_select(0, idx), y.index_select(0, idx)
            return (xb, yb) if self.device is None else (xb.to(self.device), yb.to(self.device))
        key = (b, slot)
        if key not in self._staging:
            rows = min(self.batch_size, x.shape[0])
            self._staging[key] = [torch.empty((rows,) + tuple(x.shape[1:]), dtype=x.dtype).pin_memory(),
                                  torch.empty((rows,) + tuple(y.shape[1:]), dtype=y.dtype).pin_memory(), None]
        sx, sy, ev = self._staging[key]
        if ev is not None:
            ev.synchronize()                                 # the copy that last read this staging slot has finished
        torch.index_select(x, 0, idx, out=sx[:n])
        torch.index_select(y, 0, idx, out=sy[:n])
        with torch.cuda.stream(self._stream):
            xd = sx[:n].to(self.device, non_blocking=True)
            yd = sy[:n].to(self.device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self._stream)
        self._staging[key][2] = done
        return xd, yd, done

    def __iter__(self):
        sched = self._schedule()
        if not self._cuda:
            for b, idx in sched:
                yield self._stage(b, idx, 0)
            return
        pending = self._stage(*sched[0], 0) if sched else None
        for i in range(len(sched)):
            xd, yd, done = pending
            pending = self._stage(*sched[i + 1], (i + 1) % 2) if i + 1 < len(sched) else None
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(done)
            # allocated on the side stream, consumed on the compute stream: tell the caching allocator
            xd.record_stream(cur)
            yd.record_stream(cur)
            yield xd, yd


def _loaders(rank_data, batch_size, device, seed):
    seed = shared_seed(seed)
    pairs = lambda split: [(x, y) for (x, y, _q) in rank_data.buckets[split].values()]
    # the reference shuffles BOTH loaders (attncut_dataloader.py:87-88)
    return (BatchLoader(pairs("train"), batch_size, True, device, seed),
            BatchLoader(pairs("test"), batch_size, True, device, seed + 1), rank_data)


def attncut_dataloader(retrieve_data="robust04", dataset_name="bm25", batch_size=20, device=None, base=None, seed=None):
    return _loaders(RankData(retrieve_data, dataset_name, True, base, "attncut"), batch_size, device, seed)


def choopy_dataloader(retrieve_data="robust04", dataset_name="bm25", batch_size=20, device=None, base=None, seed=None):
    return _loaders(RankData(retrieve_data, dataset_name, False, base), batch_size, device, seed)


def mtcut_dataloader(retrieve_data="mq2007", dataset_name="bm25", batch_size=20, device=None, base=None, seed=None):
    """mtcut_dataloader.py:74-90: the same layout with the statistics under mtcut/ (MQ2007: 46 columns)."""
    return _loaders(RankData(retrieve_data, dataset_name, True, base, "mtcut"), batch_size, device, seed)
