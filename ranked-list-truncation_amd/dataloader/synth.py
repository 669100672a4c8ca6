"""Write a robust04-SHAPED synthetic dataset in the reference's pickle layout (the real robust04 lists
are not redistributable and are absent from the reference repository).
Sizes default to the DRMM-TKS split of the reference: 194 train / 49 test queries, 300 documents each."""
import os
import pickle

import numpy as np


def write_synthetic_robust04(base, retrieve_data="robust04", dataset_name="drmm_tks", n_train=194, n_test=49,
                             seq_len=300, seed=20240, lengths=None, stats_width=2, stats_dir="attncut"):
    """lengths: None (every list has seq_len documents, as in the reference's files) or a tuple of list lengths cycled
    over the queries (BASELINE configs[4]: (100, 200, 300)).  The draws of a length-300 set do not depend on the new
    arguments, so the round-1 fixtures stay valid."""
    rs = np.random.RandomState(seed)
    root = os.path.join(base, retrieve_data)
    os.makedirs(os.path.join(root, stats_dir), exist_ok=True)
    gt = {}
    qid = 301
    for split, n in (("train", n_train), ("test", n_test)):
        raw, stats = {}, {}
        for i in range(n):
            q = str(qid)
            qid += 1
            s_len = seq_len if not lengths else lengths[i % len(lengths)]
            prob = 0.55 * np.exp(-np.arange(s_len) / 45.0) + 0.02
            scores = np.sort(rs.standard_normal(s_len) * 2.5 + 3.0)[::-1]
            docs = [f"FT{rs.randint(900, 999)}-{q}-{j}" for j in range(s_len)]
            raw[q] = {d: float(s) for d, s in zip(docs, scores)}
            stats[q] = rs.uniform(0, 1, (s_len, stats_width)).tolist()
            rel = rs.uniform(0, 1, s_len) < prob
            if not rel.any():
                rel[rs.randint(0, 10)] = True
            gt[q] = [d for d, r in zip(docs, rel) if r] + [f"unretrieved-{q}"]
        with open(os.path.join(root, f"{dataset_name}_{split}.pkl"), "wb") as f:
            pickle.dump(raw, f)
        with open(os.path.join(root, stats_dir, f"{dataset_name}_{split}.pkl"), "wb") as f:
            pickle.dump(stats, f)
    with open(os.path.join(root, "gt.pkl"), "wb") as f:
        pickle.dump(gt, f)
    return root
