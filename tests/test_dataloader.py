"""Data path: our loader reads the reference's pickle layout and yields the same
tensors as the reference's own loaders did on the same files (golden fixture: tools/make_golden.py data)."""
import os

import numpy as np
import pytest
import torch

import golden_util as gu


def _write(tmp_path):
    from dataloader.synth import write_synthetic_robust04
    write_synthetic_robust04(str(tmp_path), "robust04", "drmm_tks", n_train=7, n_test=3, seq_len=300, seed=77)


def test_loader_matches_reference_loader_output(tmp_path):
    from dataloader import RankData
    _write(tmp_path)
    gold = gu.load("dataloader_synth77")
    at = RankData("robust04", "drmm_tks", with_stats=True, base=str(tmp_path))
    cp = RankData("robust04", "drmm_tks", with_stats=False, base=str(tmp_path))
    for prefix, rd in (("at", at), ("cp", cp)):
        np.testing.assert_array_equal(rd.getX_train().numpy(), gold[f"{prefix}_X_train"])
        np.testing.assert_array_equal(rd.getX_test().numpy(), gold[f"{prefix}_X_test"])
        np.testing.assert_array_equal(rd.gety_train().numpy(), gold[f"{prefix}_y_train"])
        np.testing.assert_array_equal(rd.gety_test().numpy(), gold[f"{prefix}_y_test"])
    assert at.getX_train().shape == (7, 300, 3) and cp.getX_train().shape == (7, 300, 1)


def test_batches_cover_every_list_once(tmp_path):
    from dataloader import at_dataloader
    _write(tmp_path)
    train, test, data = at_dataloader("robust04", "drmm_tks", batch_size=3, base=str(tmp_path), seed=1)
    assert len(train) == 3 and len(test) == 1
    seen = torch.cat([x for x, _ in train])
    assert seen.shape == (7, 300, 3)
    ref = data.getX_train()
    # every list appears exactly once per epoch (shuffled)
    keys = sorted(float(v) for v in seen[:, :, 0].sum(1))
    assert keys == sorted(float(v) for v in ref[:, :, 0].sum(1))
    order1 = [float(v) for x, _ in train for v in x[:, 0, 0]]
    order2 = [float(v) for x, _ in train for v in x[:, 0, 0]]
    assert order1 != order2 or len(order1) < 3          # reshuffled between epochs
    for x, y in test:
        assert x.shape[0] == 3 and y.shape == (3, 300)
        assert set(np.unique(y.numpy())) <= {0.0, 1.0}


def test_length_buckets_round_robin(tmp_path):
    """BASELINE configs[4]: lists of 100 / 200 / 300 documents -> homogeneous batches, buckets served round-robin,
    every list exactly once per epoch (SURVEY.md 8f N1 'length bucketing')."""
    from dataloader import at_dataloader
    from dataloader.synth import write_synthetic_robust04
    write_synthetic_robust04(str(tmp_path), "robust04", "drmm_tks", n_train=20, n_test=6, seed=5, lengths=(100, 200, 300))
    train, test, data = at_dataloader("robust04", "drmm_tks", batch_size=3, base=str(tmp_path), seed=2)
    assert data.lengths == [100, 200, 300] and data.n_features == 3
    sizes = {s: data.buckets["train"][s][0].shape[0] for s in data.lengths}
    assert sizes == {100: 7, 200: 7, 300: 6}
    seq = [(x.shape[1], x.shape[0]) for x, _ in train]
    assert len(seq) == len(train) == 3 + 3 + 2
    assert [s for s, _ in seq[:6]] == [100, 200, 300, 100, 200, 300]            # round-robin over the buckets
    for s in data.lengths:
        got = torch.cat([x for x, _ in train if x.shape[1] == s])
        want = data.buckets["train"][s][0]
        assert sorted(float(v) for v in got[:, :, 0].sum(1)) == sorted(float(v) for v in want[:, :, 0].sum(1))
    for x, y in test:
        assert y.shape == (x.shape[0], x.shape[1])


def test_generic_statistics_width_and_mtcut_dir(tmp_path):
    """mtcut_dataloader.py:48: column_stack(score, statistics) for any statistics width (MQ2007: 46 -> 47 features)."""
    from dataloader import mc_dataloader
    from dataloader.synth import write_synthetic_robust04
    write_synthetic_robust04(str(tmp_path), "mq2007", "bm25", n_train=5, n_test=2, seq_len=40, seed=9, stats_width=46,
                             stats_dir="mtcut")
    train, _test, data = mc_dataloader("mq2007", "bm25", batch_size=4, base=str(tmp_path), seed=0)
    assert data.n_features == 47 and data.lengths == [40]
    assert [tuple(x.shape) for x, _ in train] == [(4, 40, 47), (1, 40, 47)]


def test_shard_bounds_partition():
    from rlt_hip.parallel import shard_bounds
    for n in (0, 1, 2, 7, 8, 63, 64):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_bounds(n, r, world)
                assert 0 <= lo <= hi <= n and hi - lo in (n // world, n // world + 1)
                cover.extend(range(lo, hi))
            assert cover == list(range(n))


def test_feature_width_mismatch_is_reported_at_load(tmp_path):
    """Buckets / splits whose statistics widths differ cannot feed one model: RankData says so when the files are read
    (the width is inferred per bucket from its first query)."""
    import pickle
    from dataloader import RankData
    from dataloader.synth import write_synthetic_robust04
    root = write_synthetic_robust04(str(tmp_path), "robust04", "drmm_tks", n_train=6, n_test=4, seq_len=30, seed=1)
    path = os.path.join(root, "attncut", "drmm_tks_test.pkl")
    with open(path, "rb") as f:
        stats = pickle.load(f)
    stats = {q: [row + [0.5] for row in rows] for q, rows in stats.items()}       # the test split gets a third statistic
    with open(path, "wb") as f:
        pickle.dump(stats, f)
    with pytest.raises(ValueError, match="feature width differs"):
        RankData("robust04", "drmm_tks", True, str(tmp_path))


def test_test_split_lengths_are_exposed(tmp_path):
    """run.py refuses single-length models (Choopy, MMOECut) when the test split holds a length the train split lacks;
    RankData exposes both."""
    import pickle
    from dataloader import RankData
    from dataloader.synth import write_synthetic_robust04
    root = write_synthetic_robust04(str(tmp_path), "robust04", "drmm_tks", n_train=4, n_test=3, seq_len=30, seed=2)
    path = os.path.join(root, "drmm_tks_test.pkl")
    with open(path, "rb") as f:
        raw = pickle.load(f)
    q0 = next(iter(raw))
    raw[q0] = dict(list(raw[q0].items())[:20])                                     # one test list is shorter
    with open(path, "wb") as f:
        pickle.dump(raw, f)
    data = RankData("robust04", "drmm_tks", False, str(tmp_path))
    assert data.lengths == [30] and data.test_lengths == [20, 30]
