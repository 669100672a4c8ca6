"""Data path: our loader reads the reference's pickle layout and yields the same
tensors as the reference's own loaders did on the same files (golden fixture: tools/make_golden.py data)."""
import os

import numpy as np
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
